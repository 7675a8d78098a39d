"""Build-time invariants the hand-written asm of the run kernels relies on (no GPU needed: hipcc cross-compiles).

k_run256v2 issues its output stores from inline asm with a scalar base and no wait states in front (1 % of the launch).
That is only safe while hipcc does not spill SGPRs in the kernel: a base reloaded from a spill lane (v_readlane = a VALU
write of an SGPR) must be five wait states old before a VMEM instruction reads it, and the hazard recognizer does not look
into inline asm (fused_v2_common.h, dma_tile).  k_run1024v2 does spill SGPRs and pays for the wait states."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_run256v2_has_no_register_spills(tmp_path):
    src = os.path.join(ROOT, "composable_sdr_amd", "csrc")
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", os.path.join(src, "kernels_fused_v2.hip"),
                          "-o", str(tmp_path / "v2.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", out.stderr)[1:]
    seen = 0
    for b in blocks:
        if "k_run256v2" not in b.splitlines()[0]:
            continue
        seen += 1
        sg = int(re.search(r"SGPRs Spill: (\d+)", b).group(1))
        vg = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        assert sg == 0, f"k_run256v2 spills SGPRs ({sg}): its asm stores have no wait states in front (V2_SNOP)"
        # VGPRs: the whole-band kernels must not spill at all; a shard variant may park a value across the tile loop (round 5: one
        # dword of <FM, 2>, stored in front of the loop and reloaded behind it) as long as no scratch access sits inside a loop
        whole_band = "Li1EEE" in b.splitlines()[0]
        assert vg == 0 or (not whole_band and vg <= 4), f"k_run256v2 spills VGPRs ({vg})"
    assert seen == 8                                     # <FM>, <CF32> and the interleaved-shard variants G = 2, 4, 8 of each
    asm = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", os.path.join(src, "kernels_fused_v2.hip"),
                          "-o", str(tmp_path / "v2.s")], capture_output=True, text=True, timeout=600)
    assert asm.returncode == 0, asm.stderr[-2000:]
    block = ""
    for line in open(tmp_path / "v2.s"):
        if line.startswith(".LBB") or line.startswith("_Z"):
            block = line
        if "scratch_" in line and "k_run256v2" not in block:
            assert "Loop" not in block, f"scratch access inside a loop of k_run256v2: {block.strip()} / {line.strip()}"
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", os.path.join(src, "kernels_run64_v2.hip"),
                          "-o", str(tmp_path / "r64.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    b = [x for x in re.split(r"remark: Function Name: ", out.stderr)[1:] if "k_run64v2" in x.splitlines()[0]]
    assert len(b) == 1
    assert int(re.search(r"SGPRs Spill: (\d+)", b[0]).group(1)) == 0 and int(re.search(r"VGPRs Spill: (\d+)", b[0]).group(1)) == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_run1024v3_has_no_register_spills_and_fits_one_workgroup_per_cu(tmp_path):
    """k_run1024v3 (512 threads, front / back wave roles, 128 VGPRs of staged output lines in the back waves, a 128-register window ring
    in the front waves): two waves per SIMD means 256 VGPRs, and anything hipcc spills lands in the tile loop -- a scratch load in the
    front waves' queue would wait for nothing they own, one in the back waves' queue for the tile DMA.  Its asm DMA reads scalar bases:
    no SGPR spills either (v_readlane -> VMEM hazard, fused_v2_common.h).  Static LDS: four tile buffers within a CU's 160 KiB."""
    src = os.path.join(ROOT, "composable_sdr_amd", "csrc")
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", os.path.join(src, "kernels_run1024_v3.hip"),
                          "-o", str(tmp_path / "b3.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = [b for b in re.split(r"remark: Function Name: ", out.stderr)[1:] if "k_run1024v3" in b.splitlines()[0]]
    assert len(blocks) == 2                               # <FM>, <CF32>
    for b in blocks:
        assert int(re.search(r"SGPRs Spill: (\d+)", b).group(1)) == 0 and int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0, b[:600]
        assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0
        assert int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1)) <= 160 * 1024


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_agc_spec_tm_keeps_its_block_registers_out_of_scratch(tmp_path):
    """k_agc_spec_tm's gain wave holds two blocks of 16 samples in registers (software pipeline).  As reference parameters of a
    lambda hipcc put those arrays into scratch memory (found in round 4: 256 bytes of private segment, a scratch round trip per
    sample): the kernel must have no private segment and no spills."""
    src = os.path.join(ROOT, "composable_sdr_amd", "csrc")
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", os.path.join(src, "kernels_agc_tail.hip"),
                          "-o", str(tmp_path / "agc.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = [b for b in re.split(r"remark: Function Name: ", out.stderr)[1:] if "k_agc_spec_tm" in b.splitlines()[0]]
    assert len(blocks) == 2
    for b in blocks:
        assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0, b[:600]
        assert int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_shard1024_keeps_its_window_and_taps_in_registers(tmp_path):
    """k_shard1024<FM | CF32, G> (round 6; 512 threads, one workgroup per CU = two waves per SIMD = 256 VGPRs): the 16-frame x 2-branch window
    ring (64 VGPRs) and the 28 taps + 4 phasors of a thread's two branches must stay in registers for the whole run -- a VGPR spill would put
    a scratch round trip into every step -- and the three tile buffers, the partial folds, Y, the stash and the per-row output staging must
    fit a CU's 160 KiB.  (SGPR spills are tolerated here: the DMA asm block puts seven wait states between a reloaded scalar and the
    VMEM instruction that reads it, fused_v2_common.h dma_piece.)"""
    src = os.path.join(ROOT, "composable_sdr_amd", "csrc")
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", os.path.join(src, "kernels_shard1024.hip"),
                          "-o", str(tmp_path / "s1.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = [b for b in re.split(r"remark: Function Name: ", out.stderr)[1:] if "k_shard1024I" in b.splitlines()[0]]
    assert len(blocks) == 4                               # <FM | CF32> x G = 4, 8
    for b in blocks:
        assert int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0, b[:600]
        assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0, b[:600]
        assert int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1)) <= 160 * 1024
        assert int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1)) == 2
