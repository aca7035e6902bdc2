#!/usr/bin/env python3
"""Average per launch of the rocprofv3 --pmc counters for the libmmt_hip kernels (arguments: a label for the measured
command, then one output directory per counter pass) -> JSON on stdout (the layout bench.py's pmc_traffic() reads:
profiles/rNN_pmc_<config>.json)."""
import csv
import glob
import json
import os
import sys

# logical kernel name <- (substring of the demangled name, further substrings that must ALL be present)
KERNELS = [
    # camera form (round 3: the kernels compute the cells themselves; last template argument true) before the geom form
    # (lss_ray_fwd<...>; for columns of up to 16 rows lss_ray_fwd_reg<...>; the camera form's other shapes lss_ray_fwd_blk<...>)
    ("lift_splat_forward_plan", ("lss_plan_fwd<float, ",)), ("lift_splat_forward_plan_bf16", ("lss_plan_fwd<unsigned short, ",)),
    ("depth_softmax_fwd", ("lss_plan_lookup_softmax",)),      # (the depth softmax with the calibration lookup riding in its launch)
    ("lss_plan_lookup", ("lss_plan_lookup",)), ("lss_plan_probe", ("lss_plan_probe",)), ("lss_plan_build", ("lss_plan_build",)),
    ("lift_splat_forward_camera", ("lss_ray_fwd_blk<float, ",)),
    ("lift_splat_forward_camera_bf16", ("lss_ray_fwd_blk<unsigned short, ",)),
    ("lift_splat_forward_camera", ("lss_ray_fwd", "<float, ", ", true>")),
    ("lift_splat_forward_camera_bf16", ("lss_ray_fwd", "<unsigned short, ", ", true>")),
    ("lift_splat_forward", ("lss_ray_fwd", "<float, ", ", false>")),
    ("lift_splat_forward_bf16", ("lss_ray_fwd", "<unsigned short, ", ", false>")),
    ("lss_zero_fill", ("lss_zero_fill",)),
    ("lift_splat_forward_tile", ("lss_splat_fwd_tile<float",)),
    ("lift_splat_forward_tile_bf16", ("lss_splat_fwd_tile<unsigned short",)),
    ("lift_splat_forward_chunked", ("vp_fwd_seg_gather<float", ", true>")),
    ("lift_splat_forward_chunked_bf16", ("vp_fwd_seg_gather<unsigned short", ", true>")),
    ("vp_fwd_seg_gather", ("vp_fwd_seg_gather<float", ", false>")),
    ("vp_fwd_seg_gather_bf16", ("vp_fwd_seg_gather<unsigned short", ", false>")),
    ("vp_bwd_prepare", ("vp_bwd_prepare",)),
    ("vp_bwd_rows_vec4", ("vp_bwd_rows_vec<float",)),
    ("vp_bwd_rows_bf16", ("vp_bwd_rows_vec<unsigned short",)),
    ("lift_splat_backward_column_camera", ("lss_col_bwd<float", ", true>")),
    ("lift_splat_backward_column_camera_bf16", ("lss_col_bwd<unsigned short", ", true>")),
    ("lift_splat_backward_column", ("lss_col_bwd<float", ", false>")),
    ("lift_splat_backward_column_bf16", ("lss_col_bwd<unsigned short", ", false>")),
    ("lift_splat_backward_camera", ("lss_ray_bwd<float", ", true>")),
    ("lift_splat_backward_camera_bf16", ("lss_ray_bwd<unsigned short", ", true>")),
    ("lift_splat_backward", ("lss_ray_bwd<float", ", false>")),
    ("lift_splat_backward_bf16", ("lss_ray_bwd<unsigned short", ", false>")),
    ("lift_splat_backward_tile", ("lss_splat_bwd_tile<float",)),
    ("lift_splat_backward_tile_bf16", ("lss_splat_bwd_tile<unsigned short",)),
    ("lift_splat_backward_pixel", ("lift_splat_backward_kernel<float",)),
    ("lift_splat_backward_pixel_bf16", ("lift_splat_backward_kernel<unsigned short",)),
    ("vp_planned_items", ("vp_planned_items",)), ("vp_planned_fold", ("vp_planned_fold",)),
    ("lift_kernel", ("lift_kernel<",)), ("lift_kernel_bf16", ("lift_kernel_bf16",)), ("lift_backward_vec4", ("lift_backward_vec4",)),
    ("vox_cells", ("vox_cells",)), ("vox_own", ("vox_own",)), ("vox_emit", ("vox_emit",)),
    ("fill_i32_kernel", ("fill_i32_kernel",)), ("scatter_map_kernel", ("scatter_map_kernel",)),
    ("scatter_write_nhwc_kernel", ("scatter_write_nhwc_kernel",)), ("scatter_backward_nhwc_kernel", ("scatter_backward_nhwc_kernel",)),
    ("scatter_write_strided_table_kernel", ("scatter_write_strided_table_kernel",)), ("scatter_write_strided_kernel", ("scatter_write_strided_kernel",)),
    ("scatter_backward_strided_kernel", ("scatter_backward_strided_kernel",)),
    ("depth_softmax_fwd", ("depth_softmax_fwd",)), ("depth_softmax_bwd", ("depth_softmax_bwd",)),
    ("normalize_flip_kernel", ("normalize_flip_kernel",)), ("hflip_kernel", ("hflip_kernel",)),
    ("depth_project_kernel", ("depth_project_kernel",)), ("depth_bins_kernel", ("depth_bins_kernel",)),
    ("scatter_write_nhwc_table_kernel", ("scatter_write_nhwc_table_kernel",)), ("scatter_backward_nhwc_unique_kernel", ("scatter_backward_nhwc_unique_kernel",)),
    # implicit-GEMM form of the deformable convolution (csrc/deform_conv_mfma.hip); dcn_plan_taps before the column form's dcn_plan
    ("dcn_pack_weights", ("dcn_pack_weights",)), ("dcn_fwd_mfma", ("dcn_fwd_mfma",)), ("dcn_plan_taps", ("dcn_plan_taps",)),
    ("dcn_wgrad_mfma", ("dcn_wgrad_mfma",)), ("dcn_wgrad_reduce", ("dcn_wgrad_reduce",)), ("dcn_dgrad_gather", ("dcn_dgrad_gather",)),
    ("dcn_dgrad_mfma", ("dcn_dgrad_mfma",)), ("dcn_offset_reduce_parts", ("dcn_offset_reduce_parts",)),
    ("dcn_col2im_gather", ("dcn_col2im_gather",)), ("dcn_offset_grad", ("dcn_offset_grad",)), ("dcn_plan", ("dcn_plan",)),
    ("dcn_col2im", ("dcn_col2im",)), ("dcn_im2col", ("dcn_im2col",)),
    ("bev_warp_kernel", ("bev_warp_kernel",)), ("bev_warp_backward_gather", ("bev_warp_backward_gather",)),
]


def logical(name):
    for key, subs in KERNELS:
        if all(s in name for s in subs):
            return key
    return None


label = sys.argv[1]
acc = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            k = logical(r["Kernel_Name"])
            if k is None:
                continue
            key = (k, r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])   # sum over XCD rows
        for (k, _, c), v in sorted(per_dispatch.items(), key=lambda kv: int(kv[0][1])):      # in launch order
            acc.setdefault(k, {}).setdefault(c, []).append(v)
res = {"note": "rocprofv3 --pmc, one counter group per pass (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum "
               "TCC_EA0_ATOMIC_sum TCC_HIT_sum); averages per launch over the LATER HALF of the run's launches -- the steady state: the "
               "exclusive-cell cache of the forward learns a calibration over its first three calls (`first_half` holds the earlier "
               "launches' averages where they differ) (tools/collect_profiles.sh). FETCH_SIZE / WRITE_SIZE are KiB. "
               "gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte read requests of wide "
               "(16 B/lane) loads at 64 B, so traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; the guide calls other access widths "
               "uncalibrated -- for the LiDAR kernels (4- and 8-byte scattered accesses, 64-bit atomics) read the figure as an upper "
               "bound. Cross-check: TCC_EA0_RDREQ_sum*128 B reads, TCC_EA0_WRREQ_sum*64 B writes+atomics.",
       "command": label, "kernels": {}}
for k, ctrs in sorted(acc.items()):
    def later(v):
        return v[len(v) // 2:]
    e = {c: sum(later(v)) / len(later(v)) for c, v in sorted(ctrs.items())}
    first = {c: sum(v[:len(v) // 2]) / max(1, len(v) // 2) for c, v in sorted(ctrs.items()) if len(v) >= 2}
    if any(abs(first[c] - e[c]) > 0.05 * max(abs(e[c]), 1.0) for c in first):
        e["first_half"] = first
    e["launches"] = max(len(v) for v in ctrs.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["traffic_bytes"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    if "TCC_EA0_RDREQ_sum" in e:
        e["read_bytes_RDREQ_x128"] = e["TCC_EA0_RDREQ_sum"] * 128
        e["write_bytes_WRREQ_x64"] = e["TCC_EA0_WRREQ_sum"] * 64
    if "TCC_EA0_ATOMIC_sum" in e:
        e["atomic_bytes"] = e["TCC_EA0_ATOMIC_sum"] * 64        # memory-side atomic requests of 64 B (bench.py: roofline.atomic_side)
    res["kernels"][k] = e
print(json.dumps(res, indent=1))
