"""Build-time checks of the LDS-resident pass kernels (no GPU: hipcc cross-compiles gfx950).

* Every instance `launch_lds_one` can pick keeps its registers: no VGPR spill, no scratch
  (the dominant instance spilled 6 VGPRs at the 128-register budget of a 1024-thread
  workgroup).  The report is written to profiles/r04_spmm_lds_resource_usage.txt.
* The walk of those instances waits for its stream with `s_waitcnt vmcnt(PF + 1)` (+ 1: the rows of a balanced slab) while the prefetch
  of the next slab is in flight.  That count is only right if the prefetch is EXACTLY PF 16-byte
  loads (2 PF 8-byte loads when staged element-wise) plus the one load of the slab's bnd words,
  all unconditional: checked in the ISA.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vireo_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _pf_from_header():
    """VRX_LDS_PF as vrx_kernels.h derives it from its VRX_LDS_WAVES / VRX_LDS_RING defaults
    (the expected load count must follow the header, not a constant kept here)"""
    src = open(os.path.join(CSRC, "vrx_kernels.h")).read()
    waves = int(re.search(r"#define VRX_LDS_WAVES (\d+)", src).group(1))
    ring = int(re.search(r"#define VRX_LDS_RING (\d+)", src).group(1))
    assert "(160 * 1024 - VRX_LDS_WAVES * VRX_RING * 4) / (VRX_LDS_WAVES * 64 * 16)" in src
    return (160 * 1024 - waves * ring * 4) // (waves * 64 * 16)


PF = _pf_from_header()      # 8 at 16 waves: (160 KiB - 32 KiB of rings) / (1024 threads x 16 B)


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("isa")
    asm, rep = str(d / "engine.s"), str(d / "usage.txt")
    with open(rep, "w") as err:
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "--cuda-device-only", "-S", "-o", asm, "vrx_engine.hip",
                        "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, stderr=err, check=True)
    return open(asm).read(), open(rep).read()


def _instances(report):
    out = {}
    for block in re.split(r"remark: [^\n]*Function Name: ", report)[1:]:
        name = block.split()[0]
        m = re.search(r"vrx_spmm_ldsILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        if not m:
            continue
        lpe, mode, rw, padk, split, form = (int(x) for x in m.groups())
        get = lambda key: int(re.search(key + r": (\d+)", block).group(1))      # noqa: E731
        out[name] = dict(lpe=lpe, mode=mode, rw=rw, padk=padk, split=split, form=form,
                         vgprs=get("VGPRs"), spill=get("VGPRs Spill"), sgpr_spill=get("SGPRs Spill"),
                         scratch=get(r"ScratchSize \[bytes/lane\]"), occupancy=get(r"Occupancy \[waves/SIMD\]"))
    return out


def test_ad_bd_instances_do_not_spill(isa):
    _, report = isa
    inst = _instances(report)
    hot = {k: v for k, v in inst.items() if v["form"] != 0}
    assert len(hot) >= 9          # cell pass RW 96 / 32 and variant pass x flat / 16-B-unit / element-wise staging
    lines = ["%-8s %-5s %-4s %-5s %-5s  VGPRs spill scratch occupancy" % ("form", "mode", "RW", "PADK", "SPLIT")]
    for k, v in sorted(inst.items(), key=lambda kv: (-kv[1]["form"], kv[1]["mode"], kv[1]["rw"], kv[1]["padk"], kv[1]["split"])):
        lines.append("%-8d %-5d %-4d %-5d %-5d  %5d %5d %7d %9d" % (
            v["form"], v["mode"], v["rw"], v["padk"], v["split"], v["vgprs"], v["spill"], v["scratch"], v["occupancy"]))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r04_spmm_lds_resource_usage.txt"), "w") as f:
        f.write("hipcc -Rpass-analysis=kernel-resource-usage, every vrx_spmm_lds instance "
                "(tests/test_kernel_build_cpu.py)\n" + "\n".join(lines) + "\n")
    for k, v in inst.items():        # every instance, pair-word forms included
        assert v["spill"] == 0 and v["scratch"] == 0 and v["sgpr_spill"] == 0, (k, v)
        assert v["occupancy"] >= 4, (k, v)


def test_slab_prefetch_is_a_fixed_number_of_loads(isa):
    asm, report = isa
    inst = _instances(report)
    hot = [k for k, v in inst.items() if v["form"] != 0]
    for name in hot:
        wide = inst[name]["padk"] != 1      # 16-B units; element-wise staging: two 8-B loads per unit
        body = asm[asm.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")].split("\n")
        bars = [i for i, l in enumerate(body) if re.match(r"\s*s_barrier", l)]
        assert len(bars) == 2, (name, len(bars))       # the two slab barriers of the walk
        between = body[bars[0]:bars[1]]
        unit = r"\s*global_load_dwordx4 " if wide else r"\s*global_load_dwordx2 "
        n_unit = sum(1 for l in between if re.match(unit, l))
        n_bnd = sum(1 for l in between if re.match(r"\s*global_load_dword ", l))
        others = [l for l in between if re.match(r"\s*(global|buffer|flat|scratch)_load", l)
                  and not re.match(unit, l) and not re.match(r"\s*global_load_dword ", l)]
        n_pf = PF if wide else 2 * PF
        # one 4-byte load of the slab's bnd words and a second one -- the rows the wave stages for the slab
        # behind it (balanced slabs, TiledStream::perm) -- issued whether or not the stream is balanced, so
        # that the count the walk waits with is one constant
        n_rec = 2
        # (the flat instances hold the prefetch twice -- contiguous slab or gathered through the tile's list --
        #  behind one scalar branch on the list pointer: n_pf loads execute either way)
        alt = 2 if inst[name]["padk"] in (0, 2) else 1      # (element-wise staging selects per load instead)
        assert (n_unit, n_bnd, others) == (alt * n_pf, n_rec, []), (name, n_unit, n_bnd, others)
        if alt == 2:
            units = [i for i, l in enumerate(between) if re.match(unit, l)]
            assert any(re.match(r"\s*s_c?branch", l) for l in between[units[n_pf - 1]:units[n_pf]]), name
        # the loads sit in one block the scalar branch `s + 1 < s_hi` guards: no lane-mask branch
        loads = [i for i, l in enumerate(between) if re.match(r"\s*global_load_dword", l)]
        # (EXEC is never narrowed around them: a `s_cbranch_execnz` that the uniform `perm ? ... : ...` selects
        #  of the element-wise instances leave behind always jumps -- EXEC is full -- and skips no load)
        assert not any("saveexec" in l or "s_cbranch_execz" in l for l in between[loads[0]:loads[-1] + 1]), name
        assert any("s_waitcnt vmcnt(%d)" % (n_pf + n_rec) in l for l in body), name


def test_balance_greedy_column_loop_waits_for_no_memory(isa):
    """vrx_balance_greedy (the balanced-slab greedy, one wave per (tile, block)): a strictly sequential kernel
    whose run time is its dependent-latency chain per column -- no scratch, no barrier in the loop (a
    __syncthreads() per column also waits for that column's two global stores), and no `s_waitcnt
    vmcnt` at the head of the column loop (the wait for the batch of order keys sits in front of the loop:
    left to the compiler it lands inside and waits for the previous column's stores every time)."""
    asm, report = isa
    block = [b for b in re.split(r"remark: [^\n]*Function Name: ", report)[1:] if b.startswith("_Z18vrx_balance_greedy")]
    assert len(block) == 1
    get = lambda key: int(re.search(key + r": (\d+)", block[0]).group(1))      # noqa: E731
    assert get("VGPRs Spill") == 0 and get("SGPRs Spill") == 0 and get(r"ScratchSize \[bytes/lane\]") == 0
    name = block[0].split()[0]
    body = asm[asm.index("\n" + name + ":"):]
    body = body[:body.index("s_endpgm")].split("\n")
    assert sum(1 for l in body if re.match(r"\s*s_barrier", l)) <= 1      # (one wave: the compiler may drop it)
    # the column loop = the depth-2 loop; its header block runs to the next label
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=2" in l]
    assert len(heads) == 1, heads
    nxt = next(i for i in range(heads[0] + 1, len(body)) if re.match(r"\.LBB\d+_\d+:", body[i]))
    head = body[heads[0]:nxt]
    assert not any("s_waitcnt vmcnt" in l for l in head), [l for l in head if "waitcnt" in l]
    assert any("ds_bpermute_b32" in l for l in body) and any("row_mirror" in l for l in body)
