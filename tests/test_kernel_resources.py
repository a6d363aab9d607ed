"""CPU: what the gfx950 code objects need per wave (hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled,
no GPU).  The streams are latency-bound, so occupancy is part of the design (an instantiation
that slipped from 65 to 115 VGPRs ran 40 % behind): no kernel may spill to scratch, and the hot instantiations
keep at least 6 waves per SIMD."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym-lmaze_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

HOT = {   # mangled-name fragment -> minimum waves per SIMD
    "step_shared_kernelILi11ELi0ELb1ELi32ELb1E": 7,        # C3, the metric (32 envs per workgroup, uncapped)
    "step_shared_kernelILi11ELi0ELb1ELi64ELb1E": 7,        # C3 with the fused reset (64 envs, 4 workgroups per CU)
    "step_shared_wave8_kernelILi0ELb1ELi64E": 8,           # C2 (wave-autonomous; the workgroup kernel below is kept for large 8x8 batches)
    "step_shared_kernelILi8ELi0ELb1ELi128E": 7,
    "step_perenv_wave_kernelILi32ELi0ELb1ELb1E": 6,        # C5
    "foveal_kernelILi1ELi0ELi32ELi14ELb0E": 6,             # v1 step
    "foveal_kernelILi2ELi0ELi128ELi18ELb0E": 6,            # v2 step
    # v4-v6 since round 3 (clock-relative visit map, one lane per window row): a chunk's tile rows are held in registers
    # across one barrier -- 27-45 VGPRs -- so that every load is in flight at once; measured at 1M envs, v4 303-320 us at
    # 4 waves per SIMD against 363-376 for the variants that kept 5-6.  The floor guards against a slip
    # below that.
    "foveal_kernelILi4ELi0ELi128ELi18ELb0E": 4,            # v4 step (library default: 128 envs per workgroup)
    "foveal_kernelILi4ELi0ELi64ELi18ELb0E": 4,
    "foveal_kernelILi4ELi0ELi64ELi18ELb1E": 4,             # v4 step with the reset fused in
    "foveal_kernelILi5ELi0ELi128ELi18ELb1E": 4,            # v5/v6 two-level step (reset + plannerStep + step)
    "foveal_kernelILi5ELi0ELi64ELi18ELb1E": 4,
    "render_expanded_stream_kernelILi11ELi7ELb1E": 8,
    "render_planes_stream_kernelILb1E": 8,
}


def _usage(src, tmp):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                          "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src),
                          "-o", os.path.join(tmp, src + ".o")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_scratch_and_hot_kernels_keep_their_occupancy(tmp_path):
    with ThreadPoolExecutor(3) as ex:
        parts = list(ex.map(lambda f: _usage(f, str(tmp_path)), ["lmaze_step.hip", "lmaze_aux.hip", "lmaze_foveal.hip"]))
    kernels = {k: v for p in parts for k, v in p.items()}
    assert len(kernels) > 100                                   # every instantiation of the three files
    spills = {k: v for k, v in kernels.items() if v.get("ScratchSize", 0) != 0}
    assert not spills, spills
    for frag, need in HOT.items():
        hit = [(k, v) for k, v in kernels.items() if frag in k]
        assert hit, "no instantiation matches %s" % frag
        for k, v in hit:
            assert v["Occupancy"] >= need, (k, v)
