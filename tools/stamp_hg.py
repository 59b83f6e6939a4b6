#!/usr/bin/env python3
"""Phase breakdown of the partitioned hash-grid backward (hg_partition<scatter> and hg_reduce): builds csrc/umhs_kernels.hip with
-DUMHS_HG_STAMP into tools/_alt/libumhs_hgstamp.so (CPU box: `python tools/stamp_hg.py build`), then on the GPU box runs the apply
half on the bench batch's positions (C2 / C5 sample counts) and prints, per level group, the cycles EVERY wave spent between
consecutive stamps (s_memtime pinned by scheduling barriers; "drain" stamps wait for the wave's vector-memory operations first).
The stamped build forbids the overlaps the product has: read its SHARES.  `--json DIR` also writes DIR/hg_stamps_<case>.json."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd")
ALT = os.path.join(ROOT, "tools", "_alt")
LIB = os.path.join(ALT, "libumhs_hgstamp.so")
sys.path[:0] = [ROOT, PKG]
if sys.argv[1:2] == ["build"]:
    from umhsnerf import build as B
    os.makedirs(ALT, exist_ok=True)
    B.build_lib()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(ALT, "umhs_kernels.hgstamp.o")
    subprocess.check_call([hipcc, *B.FLAGS, "-DUMHS_HG_STAMP", f"-I{B.INCLUDE}", f"-I{B.CSRC}", "-c", os.path.join(B.CSRC, "umhs_kernels.hip"), "-o", obj])
    objs = [obj]
    for src in B.SOURCES:
        if src == "umhs_kernels.hip":
            continue
        for suffix, _ in B.UNITS.get(src, (("", []),)):
            objs.append(os.path.join(B.CSRC, src.replace(".hip", suffix + ".o")))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    sys.exit(0)
import torch
from umhsnerf import _hip, ops
_hip.LIB_PATH = LIB
lib = _hip.lib()
lib.umhs_debug_hg_stamps.argtypes = [ctypes.c_void_p]
import bench
dev = torch.device("cuda", 0)
PH = {0: ["prologue: bucket prefix of the run (wave 0, DPP scan), barrier", "inputs (prefetched a run ahead) until arrived", "hash + cell key",
          "DPP run merge (8 corner sums when the wave merges)", "records + LDS placement (1 atomic + 1 b128 write per record)", "wave max", "barrier",
          "record write-out (1 b128 LDS read + 1 dwordx4 store per record) until stored"],
      1: ["prologue: counts, tile zero, barrier", "record loads (dwordx4) until arrived", "-", "corner values + fixed-point conversion + LDS 64-bit atomics until done",
          "barrier", "tile -> gradient store + Adam epilogue until stored"]}
GROUPS = [("levels 0-4", range(0, 5)), ("levels 5-9", range(5, 10)), ("levels 10-15", range(10, 16)), ("all", range(16))]
CASES = {"C2": (4096, 31), "C5": (8192, 141), "C3": (8192, 128)}
res = {}
for name in os.environ.get("CASES", "C2,C5").split(","):
    R, B = CASES[name]
    N, log2_T = R * 64, 19
    layout = ops.FieldLayout(6, 31, True, log2_T)
    fs = ops.FieldSpec(layout, 0.4, True, scalings=ops.hash_scalings().to(dev))
    b = bench.synthetic_batch(R, 64, 31, seed=42, device=dev)
    _, pos, _ = ops.positions_fwd(b["origins"], b["directions"], b["starts"].view(-1), b["ends"].view(-1), fs)
    d_enc = torch.randn(16, N, 2, device=dev) * 1e-3
    T2 = (16 << log2_T)
    table, p, m, v = (torch.zeros(T2, 2, device=dev) for _ in range(4))
    adam = dict(table=p, exp_avg=m, exp_avg_sq=v, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, step=1, level_begin=5)
    def run():
        ops.hashgrid_bwd_prepare(pos, fs.scalings, log2_T)
        ops.hashgrid_bwd_apply(pos, d_enc, fs.scalings, log2_T, table, True, overwrite=True, adam=adam)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.umhs_debug_hg_stamps_clear()
    run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    lib.umhs_debug_hg_stamps(buf)
    res[name] = {}
    for kern, kname in ((0, "hg_partition<scatter>"), (1, "hg_reduce")):
        rows = [[buf[(kern * 16 + l) * 16 + q] for q in range(16)] for l in range(16)]
        res[name][kname] = {}
        print(f"{name}: {kname}  (cycles per wave, mean over the waves of the level group; share of the wave's stamped time)")
        for gname, lv in GROUPS:
            waves = sum(rows[l][15] for l in lv)
            tot = sum(sum(rows[l][:15]) for l in lv)
            if not waves:
                continue
            ent = {"waves": waves, "cycles_per_wave": round(tot / waves, 1), "phases": {}}
            print(f"  {gname}: {waves} waves, {tot / waves:.0f} cycles per wave")
            for q, label in enumerate(PH[kern]):
                c = sum(rows[l][q] for l in lv)
                ent["phases"][label] = {"cycles_per_wave": round(c / waves, 1), "share": round(c / max(tot, 1), 4)}
                print(f"    {label:72s} {c / waves:9.0f}  {100.0 * c / max(tot, 1):5.1f} %")
            res[name][kname][gname] = ent
if "--json" in sys.argv:  # --json DIR: one file per case, DIR/hg_stamps_<case>.json
    d = sys.argv[sys.argv.index("--json") + 1]
    os.makedirs(d, exist_ok=True)
    for name, r in res.items():
        with open(os.path.join(d, f"hg_stamps_{name}.json"), "w") as f:
            json.dump({"case": name, "csrc_sha": bench.csrc_hash(), "note": "cycles per wave and run of 512 samples (scatter) / per wave (reduce) between "
                       "s_memtime stamps of the -DUMHS_HG_STAMP build, mean over the waves of the level group; 'until ...' phases drain the wave's "
                       "vector-memory operations first.  The stamped build forbids the product's overlaps: read shares, not lengths.", **r}, f, indent=1)
