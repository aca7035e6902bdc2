"""CPU: no kernel of libmmt_hip.so keeps anything in scratch (private segment) memory.  Every kernel here is sized to live in
registers and LDS; a private segment means an indexed local array or a register spill went to memory -- per-thread traffic
that no roofline figure of DESIGN.md accounts for.  (Round 4: a two-element array inside a probe struct of the fused forward
sent the struct to scratch: +18 MB of HBM traffic and +4 us per launch, found in the PMC pass, invisible to every parity
test.)  Reads the gfx950 code objects out of the built library with the ROCm LLVM tools; skipped where those are absent."""
import os
import re
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# kernels that are allowed a private segment:
#   rocprim's radix sort (third-party code behind mmt_voxel_pooling_plan_build, once per calibration);
#   lss_ray_bwd, the per-pixel ray-walk backward (not the level rig's default): __launch_bounds__(256, 6) caps it at 80 VGPRs so
#   that every workgroup of the launch is resident at once, and the camera form's geometry phase spills 12-76 bytes per
#   thread for it -- a measured trade (round 3: 36.5 -> 26.7 us with the cap), outside the walk's loop.
# * lss_plan_lookup (round 6: the probe + lss_plan_build in one launch; the scratch is the build part's) /
#   lss_plan_build (lift_splat_plan.hip): learns a calibration's plan ONCE (one 1024-thread workgroup per calibration, phases of
#   plain index arithmetic with per-thread row-cell arrays); not on the step's steady-state path.
# * lss_plan_fwd<.., 5> (lift_splat_plan.hip): capped at 128 VGPRs for four waves per SIMD; two loop-invariant values are spilled
#   (8-12 bytes per thread, one reload per unit, outside the pair loop) -- 22.8 us with the cap against 27.0 us without.
# * dcn_dgrad_gather<128> (deform_conv_mfma.hip): eleven waves per workgroup = three per SIMD = 168 VGPRs, of which 64 hold the wave's
#   grad_out fragments for all nine taps; two loop-carried dwordx2 values (20 bytes per thread) are stored once and reloaded once
#   per tap, outside the MFMA block (rounds of 8 waves at 256 VGPRs would waste a quarter of the wave slots at 16 x 44 pixels).
ALLOWED = ("rocprim", "lss_ray_bwd", "lss_plan_build", "lss_plan_lookup", "lss_plan_fwd", "dcn_dgrad_gather")


def _kernels(lib_path):
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path, os.path.join(tmp, "copy.so")],
                              stderr=subprocess.DEVNULL)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        out = {}
        for i, s in enumerate(starts):
            part = os.path.join(tmp, f"b{i}.bin")
            with open(part, "wb") as f:
                f.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"b{i}.co")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            name = None
            for line in notes.splitlines():
                m = re.match(r"\s+\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
                if m and name is not None:
                    out[name] = int(m.group(1))
        return out


def test_no_kernel_uses_scratch_memory(mmt_lib):
    if not all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("ROCm LLVM tools not found")
    ks = _kernels(mmt_lib.LIB_PATH)
    assert len(ks) > 60, f"only {len(ks)} kernels found in the library's code objects"
    assert any("lss_ray_fwd_reg" in k for k in ks) and any("depth_softmax_fwd" in k for k in ks)
    bad = {k: v for k, v in ks.items() if v != 0 and not any(a in k for a in ALLOWED)}
    assert not bad, f"kernels with a private (scratch) segment: {bad}"
