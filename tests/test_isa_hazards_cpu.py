"""Static checks of the zipped field backward (csrc/umhs_field_zip.h, umhs_zip_plan.h) that need no GPU:
  * the compile-time plans close (every off-chain operation placed, only AGPR-resident tiles carried into the next tile) -- printed by
    tools/zip_plan_dump.cpp built with the host compiler;
  * no inline-asm MFMA of the compiled library reads a VGPR that a VALU instruction wrote fewer than two wait states earlier
    (tools/isa_hazards.py: hipcc does not look inside inline asm; the first version of the zipped kernel computed wrong gradients
    exactly this way, with a v_accvgpr_read reload right in front of a dW product)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unsupervised-hyperspectral-nerf_amd", "csrc")


def test_zip_plans_close(tmp_path):
    exe = tmp_path / "zip_plan_dump"
    subprocess.check_call(["g++", "-std=c++17", f"-I{CSRC}", os.path.join(ROOT, "tools", "zip_plan_dump.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    heads = [l for l in out.splitlines() if l.startswith("part ")]
    assert len(heads) == 2 and all("ok=1" in h for h in heads), heads
    for h in heads:  # every slot of a tile can take one MFMA: the carried ones must fit in front of the next tile's own
        carried = int(h.split("carried")[1].split(";")[0])
        slots = int(h.split(" slots")[0].split(",")[-1])
        assert 0 < carried < slots // 4, h


def test_no_inline_asm_mfma_reads_a_freshly_written_vgpr(built_library):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_hazards

    objs = [os.path.join(CSRC, f"umhs_field{u}.o") for u in ("", "_p0", "_p1", "_p0f")]  # the file's four translation units (umhsnerf/build.py)
    assert all(os.path.exists(o) for o in objs), "run __graft_entry__.build() first"
    text = "".join(isa_hazards.disassemble(o) for o in objs)
    assert text.count("v_mfma_f32_16x16x16_bf16 a[") > 1000, "the dW products of the transpose-free backward were not found in the disassembly"
    stats = {}
    bad = isa_hazards.check(text, stats)
    assert not bad, bad[:5]
    # both operand files were seen and checked: the products on VGPR tiles and the carried ones whose tiles live in AGPRs
    assert stats.get("v", 0) > 1000 and stats.get("a", 0) > 500, stats
