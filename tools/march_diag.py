"""The marcher alone on a batch of the bench's sampler scene (GPU box): one-thread-per-ray walk + emission (UMHS_MARCH_SERIAL form) against
umhs_march_walk + replay, count-only and with the scratch rows, rays per wave of the emission kernel 16 .. 1; REP=8 repeats the 4,096
rays to 32,768.  ALT=path/to/lib.so times an ablation build (only the timed calls use it, never the scene)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from umhsnerf import sampler as smp, _hip
from umhsnerf._hip import ptr
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["C2"]
cap = {}
orig = smp.march_begin
def spy(*a, **k):
    cap["a"], cap["k"] = a, k
    return orig(*a, **k)
smp.march_begin = spy
pipe, c2w = bench.sampler_scene(cfg, dev, warm=int(os.environ.get("WARM", "300")))
for s in range(300, 304): pipe.get_train_loss_dict(s)
torch.cuda.synchronize()
if os.environ.get("EVAL"):  # the rays of an eval-image chunk (32,768 consecutive pixels of a 256 x 256 camera) instead of a training batch
    from umhsnerf.data.umhs_dataparser import Cameras
    from umhsnerf.data.umhs_datamanager import ResidentSplit
    H = 256
    f = 30.0 * H / 64.0
    cams = Cameras(c2w[:1].contiguous(), torch.full((1,), f), torch.full((1,), f), torch.full((1,), H / 2), torch.full((1,), H / 2), H, H)
    split = ResidentSplit(cams, torch.zeros(1, H, H, 3), torch.zeros(1, H, H, cfg["B"]), dev)
    with torch.no_grad():
        pipe.model.eval()
        pipe.model.get_outputs_for_camera_ray_bundle(split.image_rays(0))
    torch.cuda.synchronize()
a, k = cap["a"], cap["k"]
names = ["origins", "directions", "bin", "roi", "levels", "res", "near", "far", "step", "cone", "nears", "fars", "jitter", "jitter_step"]
args = dict(zip(names, a)); args.update(k)
o, d = _hip.f32c(args["origins"]), _hip.f32c(args["directions"])
REP = int(os.environ.get("REP", "1"))
o, d = o.repeat(REP, 1).contiguous(), d.repeat(REP, 1).contiguous()
for kk in ("nears", "fars", "jitter"):
    if args.get(kk) is not None:
        args[kk] = args[kk].repeat(REP)
R = o.shape[0]
print("R", R, {n: args.get(n) for n in ["levels", "res", "near", "far", "step", "cone", "jitter_step"]}, "roi", [float(v) for v in args["roi"]])
roi = (C.c_float * 6)(*[float(v) for v in args["roi"]])
jit = _hip.f32c(args["jitter"]) if args.get("jitter") is not None else None
nears = _hip.f32c(args["nears"]) if args.get("nears") is not None else None
fars = _hip.f32c(args["fars"]) if args.get("fars") is not None else None
WS = None
def run(lib, reps=1, walk=False):
    global WS
    if walk:
        WS = torch.empty(lib.umhs_march_walk_workspace_bytes(R), device=dev, dtype=torch.uint8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(reps):
            e0.record()
            rc = lib.umhs_march_walk(ptr(o), ptr(d), R, ptr(args["bin"]), roi, args["levels"], args["res"], args["near"], args["far"], ptr(nears), ptr(fars), ptr(jit), float(args.get("jitter_step", 0.0)), ptr(WS), WS.numel(), _hip.stream())
            assert rc == 0
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print("march_walk %.3f ms" % min(ts))
        al = lambda x: (x + 255) & ~255
        vcap = int(os.environ.get("UMHS_MARCH_VCAP", "512"))
        base = (-WS.data_ptr()) % 256
        off_n = base + 2 * al(R * vcap * 4) + al(R * vcap)
        n_ent = WS[off_n:off_n + 4 * R].view(torch.int32).cpu().numpy()
        n_vox = WS[off_n + al(4 * R):off_n + al(4 * R) + 4 * R].view(torch.int32).cpu().numpy()
        print("lists: entries (runs of occupied voxels) per ray mean %.1f p90 %.0f max %d; voxels walked per ray mean %.1f max %d" % (n_ent.mean(), np.percentile(n_ent, 90), n_ent.max(), n_vox.mean(), n_vox.max()))
    else:
        WS = None
    return run1(lib, reps)
def run1(lib, reps=1):
    counts = torch.empty(R, device=dev, dtype=torch.int64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        rc = lib.umhs_march_count(ptr(o), ptr(d), R, ptr(args["bin"]), roi, args["levels"], args["res"], args["near"], args["far"], args["step"], args["cone"],
                                  ptr(nears), ptr(fars), ptr(jit), float(args.get("jitter_step", 0.0)), ptr(counts), ptr(WS), (WS.numel() if WS is not None else 0), _hip.stream())
        assert rc == 0
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return counts.cpu().numpy(), min(ts)
lib = _hip.lib()
if os.environ.get("ALT"):  # an ablation build: ONLY for the timed march calls below, never for the scene
    alt = C.CDLL(os.path.join(ROOT, os.environ["ALT"]))
    for n in ("umhs_march_count", "umhs_march_walk", "umhs_march_walk_workspace_bytes"):
        getattr(alt, n).restype = getattr(lib, n).restype
        getattr(alt, n).argtypes = getattr(lib, n).argtypes
    lib = alt
cnt0, t = run(lib, 5)
print("serial: march_count (no stores) %.3f ms" % t)
for rpw in (16, 8, 4, 2, 1):
    os.environ["UMHS_MARCH_RPW"] = str(rpw)
    cnt, t = run1(lib, 5) if rpw != 16 else run(lib, 5, walk=True)
    assert os.environ.get("ALT") or (cnt == cnt0).all()
    print("rpw", rpw, "walked: march_count %.3f ms" % t)
    WS0 = WS; WS = None
    cnt, t = run1(lib, 3); WS = WS0
    print("rpw", rpw, "serial: march_count %.3f ms" % t)
del os.environ["UMHS_MARCH_RPW"]
WS = WS0
cap = 1024
s0 = torch.empty(R * cap, device=dev); s1 = torch.empty(R * cap, device=dev); counts = torch.empty(R, device=dev, dtype=torch.int64)
for label, ws, rpw in (("walked", WS, 0), ("serial", None, 0), ("walked", WS, 16), ("walked", WS, 8), ("walked", WS, 4), ("walked", WS, 2), ("walked", WS, 1)):
    if rpw:
        os.environ["UMHS_MARCH_RPW"] = str(rpw)
        label += " rpw %d" % rpw
    else:
        os.environ.pop("UMHS_MARCH_RPW", None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(4):
        e0.record()
        rc = lib.umhs_march_scratch(ptr(o), ptr(d), R, ptr(args["bin"]), roi, args["levels"], args["res"], args["near"], args["far"], args["step"], args["cone"],
                                    ptr(nears), ptr(fars), ptr(jit), float(args.get("jitter_step", 0.0)), cap, ptr(counts), ptr(s0), ptr(s1), ptr(ws), (ws.numel() if ws is not None else 0), _hip.stream())
        assert rc == 0
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(label, "march_scratch (with stores, cap 1024) %.3f ms" % min(ts))
print("walked: march_count (no stores) %.3f ms; samples per ray mean %.1f max %d total %d" % (t, cnt.mean(), cnt.max(), cnt.sum()))
