#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X BEV-fusion hot path.

    python bench.py --gpus N --steps K --warmup W

One rank per GPU, pure data parallel over the batch axis ("weak" scaling: the per-GPU batch is fixed).
With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this script starts its own N rank
processes BEFORE touching the GPU and relays rank 0's JSON line (what Lightning's launcher does for
the reference, exps/mm_training_aim.py:595-612); under torchrun it is one of the ranks.

Default (--mode train): a step = one full training step of BASELINE.json configs[3], the workload the
metric is quoted on (camera + 40k-point LiDAR fusion, bs=4/GPU, fp32) on synthetic frames resident in
HBM.  The hot-path kernels inside the timed steps carry HIP events attached to their dispatches:
`roofline` / `roofline_backward` (the camera -> BEV pooling kernels the step runs), `roofline_lidar`
(voxelize + mean + pillar scatter).  After the timed region rank 0 also times the DROP-IN
voxel_pooling op at the same shape and geometry (`roofline_voxel_pooling[_backward]`,
`hotpath_samples_per_s`) -- the like-for-like GPU figure beside `cpu_baseline`.
--mode hotpath times only voxel_pooling forward+backward (any config's camera half, fp32 or bf16
feature storage).  See DESIGN.md "Measurement".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth, same guide ("L2 (per XCD)")
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32-input MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4): same guide, "Matrix cores" (= the fp32 vector peak)
ATOMIC_PEAK_GBS = 1300.0  # chip-wide memory-side fp32 atomic-add rate, same guide ("Global float atomics": 1.26-1.36 TB/s of added bytes)

WORKLOADS = {
    "cfg2": "BASELINE configs[1]: camera-only BEVDepth (ResNet-50, 6 cams 256x704, D=112, C=80, BEV 128x128) "
            "full training step (fwd + det/depth loss + bwd + clip + AdamW)",
    "cfg3": "BASELINE configs[2]: LiDAR-only pillar path, 40k pts, 0.2 m voxels, full training step",
    "cfg3n": "BASELINE configs[2] on the reference's native LiDAR range [-204.8,-25.6,-5,204.8,25.6,3] (2048x256 pillars at 0.2 m, "
             "exps/conf_aim.py:16-18): LiDAR-only pillar path, 40k pts, full training step",
    "cfg4": "BASELINE configs[3]: LiDAR+camera fusion -- BEVDepth (ResNet-50, 6 cams 256x704, D=112, C=80, BEV 128x128) + "
            "40k-point LiDAR frames (0.2 m pillars, 512x512 canvas) concatenated in BEV; full training step "
            "(depth labels + fwd + det/depth loss + bwd + clip + AdamW)",
    "cfg5": "BASELINE configs[4]: LiDAR+radar+camera, 6 cams 512x1408, 80k pts (8 columns); bf16 storage on the hot path",
    "aim": "the reference's native configuration (exps/conf_aim.py + exps/configs/lidar_cam.py; not a BASELINE config, SURVEY 8 'for fidelity'): "
           "2 cams 704x1280, D=409, C=80, camera BEV 512x64 (0.8 m), 40k-point LiDAR frames on 2048x256 pillars, bs 4; full training step",
    "tiny": "tiny smoke configuration",
}
# camera halves for --mode hotpath (SURVEY.md section 8 shape table)
HOTPATH_SHAPES = {
    "cfg2": dict(batch=4, num_cams=6, final_dim=(256, 704), downsample=16, d_bound=(2.0, 58.0, 0.5), channels=80),
    "cfg5": dict(batch=2, num_cams=6, final_dim=(512, 1408), downsample=16, d_bound=(2.0, 58.0, 0.5), channels=80),
    "tiny": dict(batch=2, num_cams=2, final_dim=(64, 192), downsample=16, d_bound=(2.0, 58.0, 4.0), channels=16),
}
HOTPATH_SHAPES["cfg4"] = HOTPATH_SHAPES["cfg2"]
BEV_BOUNDS = dict(x_bound=(-51.2, 51.2, 0.8), y_bound=(-51.2, 51.2, 0.8), z_bound=(-5.0, 3.0, 8.0))

torch = None
dist = None
SHIPPED_MIOPEN_DB = False


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "hotpath"],
                    help="train: full data-parallel training step of the BASELINE config (default); "
                         "hotpath: only voxel_pooling forward+backward at the config's camera shape")
    ap.add_argument("--config", default="cfg4",
                    help="cfg4 (BASELINE configs[3], camera+LiDAR fusion: the workload the metric is quoted on, default) | "
                         "cfg2 | cfg3 | cfg5 | tiny")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="feature storage type of the hot-path ops (accumulation is always fp32); default f32, bf16 for cfg5")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: launcher rehearsal only (gloo, the dense detection head; the HIP hot path has no CPU form)")
    ap.add_argument("--miopen-tune", action="store_true", help="exhaustive MIOpen search (minutes of warm-up)")
    ap.add_argument("--cached-plan", action="store_true",
                    help="train mode: pass a calibration id so the unfused voxel_pooling reuses a cached sort (SURVEY 8/f3)")
    ap.add_argument("--unfused", action="store_true",
                    help="camera branch runs the reference's op sequence lift -> drop-in voxel_pooling instead of the fused "
                         "lift-splat kernels (SURVEY 8/f1, the default)")
    ap.add_argument("--fused-lift-splat", action="store_true", help="(default since round 2; accepted for old command lines)")
    ap.add_argument("--lift-splat-backward", default="auto", choices=["auto", "ray", "column"],
                    help="backward kernel of the fused camera path (auto: column on a level rig, decided once from the geometry)")
    ap.add_argument("--geom-form", action="store_true",
                    help="fused camera path: write a geom tensor with mmt_frustum_geometry every step and feed the geom form of the "
                         "kernels (the round-2 path) instead of the camera form, whose kernels compute the cells themselves")
    ap.add_argument("--rig", default="analytic", help="camera rig of the synthetic batches: analytic (the level 6-camera fan SURVEY 8d prescribes; "
                    "default) | pitched:<deg> (every camera pitched about its own x axis) | nuscenes (the reference fixture's real calibration, "
                    "cameras that are not level: tests/golden/nusc_rig.npz)")
    ap.add_argument("--calibration-ids", action="store_true",
                    help="give every synthetic batch a mats_dict['calibration_id'] (a loader that knows its rig): camera matrices, "
                         "the geometry's column summary and the backward-kernel choice are then cached per calibration")
    ap.add_argument("--full-lidar-canvas", action="store_true",
                    help="fusion configs: the reference's op sequence for the LiDAR half (full-resolution pillar canvas, nearest resize, "
                         "slice copy) instead of scattering only the sampled cells straight into the camera|LiDAR buffer")
    ap.add_argument("--aten-softmax", action="store_true",
                    help="camera branch: ATen's softmax (+ torch.where for the depth oracle) instead of mmt_depth_softmax_* (A/B)")
    ap.add_argument("--no-augment", action="store_true", help="skip augment_images (the reference's training_step always runs it)")
    ap.add_argument("--no-depth-oracle", action="store_true",
                    help="do not hand the depth labels to the model as its depth oracle (the reference does when use_depth_loss is set)")
    ap.add_argument("--conv-overlap", default=None, choices=["off", "inline", "pair", "deferred"],
                    help="weight gradients of the convolutions on a side HIP stream (ops/conv_overlap.py); default: MMT_CONV_OVERLAP or the TrainStep default")
    ap.add_argument("--head-streams", type=int, default=None,
                    help="HIP streams the CenterPoint task heads are dealt to in training (default: MMT_HEAD_STREAMS or 2; 0 = the caller's stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hotpath-leg", action="store_true", help="train mode: skip the drop-in voxel_pooling timing after the steps")
    ap.add_argument("--shared-gpu-rehearsal", action="store_true",
                    help="functional rehearsal of the N > 1 path on a box with FEWER GPUs than ranks: the ranks share devices and exchange "
                         "over gloo (RCCL refuses two ranks per device).  Without this flag such a launch fails in the preflight -- a scaling "
                         "run must never degrade silently to a host-staged backend on shared cards")
    ap.add_argument("--hotpath-leg", action="store_true",
                    help="train mode with --gpus > 1: run rank 0's drop-in voxel_pooling timing after the steps anyway (by default only a "
                         "1-GPU run does: the other ranks would sit in the final barrier for its seconds)")
    ap.add_argument("--algo", type=int, default=0, help="voxel_pooling forward algorithm flag (hotpath mode)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# rank processes

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _visible_gpus():
    """GPUs this launch can hand to ranks, WITHOUT initialising the GPU in the launcher (it must stay able to start rank
    processes): the KFD topology lists one node per agent, GPUs are those with a non-zero simd_count; ROCR / HIP visibility
    masks narrow it.  Falls back to "at least n" (= a real multi-GPU launch) when the topology cannot be read."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        count = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        count += 1
        return count if count > 0 else 1 << 30
    except OSError:
        return 1 << 30


def spawn_ranks(n, argv):
    """Start n fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and wait.
    The parent has made no GPU call (torch is not even imported yet); nothing is re-exec'd.  Rank 0's stdout is
    relayed; the exit code is non-zero if any rank failed (the others are then terminated by PID)."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(_free_port()))
    # HSA_ENABLE_IPC_MODE_LEGACY is NOT touched for a real multi-GPU launch: the image exports the value its host driver needs
    # (dmabuf IPC, which RCCL's intra-node P2P rides on) and the ranks inherit it.  Only the rehearsal in which several ranks
    # share ONE card (world > visible devices: gloo, see choose_backend) pins it to 0, because those ranks exchange CUDA
    # tensors' IPC handles on one device (tests/test_bench_ranks_gpu.py) -- DESIGN section 6.
    if n > _visible_gpus():
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # ... and keeps every rank on ONE stream: processes that share a card are time-sliced per hardware queue, and the task
        # heads' two extra queues per rank only add slices (88 s instead of ~40 for 13 cfg2 steps of two ranks)
        env.setdefault("MMT_HEAD_STREAMS", "0")
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                p = procs[r]
                if r == 0:
                    try:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                elif p.poll() is None:
                    continue
                pending.discard(r)
                if p.returncode != 0:
                    rc = rc or p.returncode or 1
                    print(f"[bench] rank {r} exited with code {p.returncode}", file=sys.stderr)
            if rc:
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                if rc:
                    p.terminate()
                try:
                    p.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    p.kill()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def use_shipped_miopen_db():
    """MIOpen's per-shape solver choice for the dense nets: mm_training_amd/miopen_db (see its README and `enable`)."""
    from mm_training_amd.miopen_db import enable
    return enable()


def _late_imports(device):
    global torch, dist, SHIPPED_MIOPEN_DB
    if device == "cuda":
        SHIPPED_MIOPEN_DB = use_shipped_miopen_db()
    import torch as _torch
    import torch.distributed as _dist
    torch, dist = _torch, _dist


def choose_backend(world, local_rank, ndev, requested="nccl"):
    """Backend and device index of a rank.  One GPU per rank -> RCCL ("nccl").  Fewer GPUs than ranks
    (functional rehearsal of the N>1 path on a 1-GPU box only): ranks share devices, which RCCL refuses, so
    EVERY rank falls back to gloo -- the decision depends on (world, ndev) alone, never on the rank, or
    rank 0 would pick RCCL while the others pick gloo and the rendezvous would hang."""
    if world > max(ndev, 1):
        return "gloo", local_rank % max(ndev, 1)
    return requested, local_rank


class _StdoutToStderr:
    """File descriptor 1 -> 2 for the duration: this image's RCCL prints a five-line version banner to STDOUT when its first
    communicator is created, and rank 0's stdout is the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init_dist(n_gpus, device="cuda", shared_gpu_rehearsal=False):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == n_gpus, f"--gpus {n_gpus} but WORLD_SIZE={world}"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if device == "cpu":
        if world > 1:
            with _StdoutToStderr():
                dist.init_process_group(backend="gloo")
                dist.barrier()
        return rank, local_rank, world
    ndev = torch.cuda.device_count()
    if world > max(ndev, 1) and not shared_gpu_rehearsal:
        # fail fast, before any communicator exists (nothing to hang on, nothing re-exec'd): every rank sees the same (world, ndev)
        if rank == 0:
            print(f"[bench] preflight FAILED: --gpus {world} but {ndev} GPU(s) visible: RCCL needs a GPU per rank "
                  "(--shared-gpu-rehearsal runs the ranks on shared cards over gloo, for functional tests only)", file=sys.stderr, flush=True)
        sys.exit(4)
    backend, local_rank = choose_backend(world, local_rank, ndev, os.environ.get("MMT_DIST_BACKEND", "nccl"))   # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            with _StdoutToStderr():                      # (device_id given: the communicator is created here, eagerly)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()
        else:
            with _StdoutToStderr():                      # (gloo announces its connections on stdout, too)
                dist.init_process_group(backend=backend)
                dist.barrier()
        if world <= ndev and "MMT_DIST_BACKEND" not in os.environ:
            # one GPU per rank: the gradient all-reduce must ride RCCL (xGMI), never silently a host-staged backend
            assert dist.get_backend() == "nccl", f"world {world} on {ndev} GPUs initialised backend {dist.get_backend()!r}, expected RCCL"
        preflight(rank, local_rank, world, ndev)
    return rank, local_rank, world


RCCL_RANKS = None      # ranks that answered the preflight all-reduce on the RCCL communicator (None: not an RCCL run)


def preflight(rank, local_rank, world, ndev):
    """Before any warm-up step of an N > 1 run: one all-reduce of ones over the communicator the gradients will use (its size must
    be the world), every rank's device gathered, and rank 0 says on stderr what the run is on.  An RCCL run whose ranks do not
    each have a GPU of their own -- fewer visible devices than ranks, or two ranks on one PCI function -- ends HERE with a
    non-zero exit on every rank (nothing is re-exec'd): RCCL would hang or crawl, and the scaling figure would mean nothing."""
    global RCCL_RANKS
    backend = dist.get_backend()
    ones = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    answered = int(ones.item())
    props = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": rank, "device": f"cuda:{local_rank}",
            "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if backend == "nccl":
        RCCL_RANKS = answered
    shared = len({g["pci_bus_id"] for g in gathered}) < world
    if rank == 0:
        print(f"[bench] preflight: backend {backend} ({'RCCL' if backend == 'nccl' else 'host-staged'}), communicator of {answered} ranks "
              f"(world {world}), {ndev} visible GPUs, HSA_ENABLE_IPC_MODE_LEGACY={mine['HSA_ENABLE_IPC_MODE_LEGACY']}", file=sys.stderr)
        for g in gathered:
            print(f"[bench] preflight:   rank {g['rank']} -> {g['device']} ({g['pci_bus_id']})", file=sys.stderr)
        sys.stderr.flush()
    bad = None
    if answered != world:
        bad = f"the communicator answered with {answered} ranks, the world is {world}"
    elif backend == "nccl" and (world > ndev or shared):
        bad = f"RCCL needs a GPU per rank: world {world}, {ndev} visible GPUs" + (", two ranks on one device" if shared else "")
    if bad:
        if rank == 0:
            print(f"[bench] preflight FAILED: {bad}", file=sys.stderr, flush=True)
        dist.destroy_process_group()
        sys.exit(4)


def distributed_info(world, local_rank, ts=None):
    """What the N > 1 path actually ran on, for the bench line (every rank calls this: it gathers): backend, world, each
    rank's device and PCI bus id, the RCCL version, and DistributedDataParallel's bucket layout / gradient bytes."""
    info = {"world": world, "backend": dist.get_backend() if world > 1 and dist.is_initialized() else None}
    props = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": int(os.environ.get("RANK", "0")), "device": f"cuda:{local_rank}", "name": props.name,
            "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        info["ranks"] = gathered
        info["ranks_share_a_device"] = len({r["pci_bus_id"] for r in gathered}) < world
    else:
        info["ranks"] = [mine]
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        info["rccl_version"] = None
    info["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    info["rccl_ranks"] = RCCL_RANKS          # ranks on the RCCL communicator (preflight all-reduce); None = no RCCL in this run
    if ts is not None:
        params = [p for n, p in ts.model.named_parameters() if p.requires_grad and ".context_se." not in n]
        info["gradient_bytes"] = int(sum(p.numel() * p.element_size() for p in params))
        ddp = ts.net if isinstance(ts.net, torch.nn.parallel.DistributedDataParallel) else None
        info["ddp"] = None
        info["reducer"] = ts.reducer.describe() if getattr(ts, "reducer", None) is not None else None
        if ddp is not None:
            cap = int(ddp.bucket_bytes_cap)
            info["ddp"] = {"bucket_cap_mb": cap / (1024 * 1024), "static_graph": bool(getattr(ddp, "static_graph", False)),
                           "gradient_as_bucket_view": bool(ddp.gradient_as_bucket_view), "broadcast_buffers": bool(ddp.broadcast_buffers),
                           "find_unused_parameters": bool(ddp.find_unused_parameters),
                           "buckets_estimate": max(1, -(-info["gradient_bytes"] // cap)),
                           "ignored_parameters": sorted(getattr(ddp, "parameters_to_ignore", []))[:4]}
    return info


def barrier(world, device="cuda"):
    if world > 1:
        dist.barrier()
    if device == "cuda":
        torch.cuda.synchronize()


def max_over_ranks(elapsed, world, device="cuda"):
    if world == 1:
        return elapsed
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------
# algorithmic bytes (BASELINE.md section 2 / SURVEY.md section 8d) and roofline objects

def algorithmic_bytes(BP, K, C, B, ny, nx, feat_bytes=4):
    """Drop-in voxel_pooling: forward 12BP geom + 12BP pos_memo + fb*C*K kept rows + 4*C*B*ny*nx BEV;
    backward 12BP pos_memo + 4*C*B*ny*nx grad_out + fb*C*BP grad_in.  fb = 4 (fp32) or 2 (bf16 storage)."""
    fwd = 12 * BP + 12 * BP + feat_bytes * C * K + 4 * C * B * ny * nx
    bwd = 12 * BP + 4 * C * B * ny * nx + feat_bytes * C * BP
    return fwd, bwd


def lift_splat_bytes(BP, K, C, B, BN_HW, ny, nx, feat_bytes=4, pos_memo=False, camera_form=False):
    """Fused lift-splat (row f1): the [BP, C] feature matrix does not exist.
    forward  12BP geom (+ 12BP pos_memo, first-generation kernels only) + fb*BP depth + fb*C*BN*HW context + 4*C*B*ny*nx BEV
    backward 12BP geom (or pos_memo) + fb*BP depth + fb*C*BN*HW context + 4*C*B*ny*nx grad_out + fb*BP grad_depth
             + fb*C*BN*HW grad_context
    The frustum-tile kernels redo the kept test from geom, so no pos_memo is written or read.
    camera_form: no geom tensor at all -- the kernels compute the cells from B*N matrices; what replaces the 12BP is the
    column summary, 8 bytes per block of 16 points (BP / 2), written by the forward and read by the backward.
    The point-wise L2-side figure adds what a point-wise gather moves: one C-row (context / grad_out) per kept point."""
    gbytes = BP // 2 if camera_form else 12 * BP
    fwd = gbytes + (12 * BP if pos_memo else 0) + feat_bytes * BP + feat_bytes * C * BN_HW + 4 * C * B * ny * nx
    bwd = gbytes + 2 * feat_bytes * BP + 2 * feat_bytes * C * BN_HW + 4 * C * B * ny * nx
    l2_fwd = fwd + K * (feat_bytes * C)
    l2_bwd = bwd + K * (4 * C)
    return fwd, bwd, l2_fwd, l2_bwd


def lidar_bytes(F, total_points, M, nf, C, B, ny, nx, voxels_T=0, sampled=None):
    """voxelize+mean: 4*F*sum(Ni) points + 16*M coors + 4*M num_points + 4*nf*M means (+ 4*T*F*M when the padded voxel
    tensor is materialised); scatter: 4*C*M + 16*M + 4*C*B*ny*nx; scatter backward: 8*C*M + 16*M.
    sampled = (M_s, oh, ow): the scatter at the fusion layer's resolution (round 4) -- only the M_s voxels on sampled cells
    are read, the output is [B, oh, ow, C] and the per-cell lookup is one 8-byte table entry per OUTPUT cell:
    scatter 4*C*M_s + 8*B*oh*ow + 4*C*B*oh*ow; backward 16*M coors + 4*C*M_s gradient rows + 4*C*M rows written."""
    vox = 4 * F * total_points + 20 * M + 4 * nf * M + 4 * voxels_T * F * M
    if sampled is not None:
        ms, oh, ow = sampled
        return vox, 4 * C * ms + 8 * B * oh * ow + 4 * C * B * oh * ow, 16 * M + 4 * C * ms + 4 * C * M
    scat = 4 * C * M + 16 * M + 4 * C * B * ny * nx
    scat_bwd = 8 * C * M + 16 * M
    return vox, scat, scat_bwd


PMC_RIG = "analytic"      # set by train_main: the camera rig of the run (the summaries are per configuration AND rig)


def pmc_traffic(config, kernels, rig=None):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary OF THIS CONFIGURATION AND CAMERA RIG (separate FETCH_SIZE /
    WRITE_SIZE passes with the gfx950 correction; tools/collect_profiles.sh writes profiles/rNN_pmc_<config>[_<rig>].json, no
    suffix = the analytic level rig).  PMC collection cannot run inside this process; None when no summary of that
    configuration and rig exists, or the newest one lacks a kernel (it was not measured -- never "as on another rig")."""
    rig = PMC_RIG if rig is None else rig
    tag = config if rig == "analytic" else f"{config}_{rig.replace(':', '_')}"
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{tag}.json")
        if not os.path.exists(path):
            continue                  # an older round's summary stands in only while this round has none for the configuration
        try:
            k = json.load(open(path))["kernels"]
            return float(sum(k[name]["traffic_bytes"] for name in kernels))
        except Exception:
            return None               # the newest summary lacks the kernel: it was not measured, not "as in an older round"
    return None


def single_stream_leg(ts, batches, steps, _lib):
    """A few more steps (after the timed region, outside every reported rate) with the whole step on ONE stream: the
    dispatch-attached events of a kernel that shares the card with the weight-gradient stream's queue measure the time it
    shared, not the kernel (the pooling backward: 110 us in the step, 27 us alone and under rocprofv3)."""
    from mm_training_amd.ops.conv_overlap import OverlapConv2d
    convs = [m for m in ts.model.modules() if isinstance(m, OverlapConv2d)]
    saved, head_streams = [m._mmt_overlap_mode for m in convs], ts.model.head.task_streams
    for m in convs:
        m._mmt_overlap_mode = "inline"
    ts.model.head.task_streams = 0
    torch.cuda.synchronize()
    _lib.TIMING = {}
    try:
        for i in range(steps):
            ts(batches[i % len(batches)])
        torch.cuda.synchronize()
    finally:
        alone, _lib.TIMING = _lib.TIMING, None
        for m, mode in zip(convs, saved):
            m._mmt_overlap_mode = mode
        ts.model.head.task_streams = head_streams
    return alone


def add_alone(entry, alone, kinds, _lib):
    """entry['alone']: the same kernels' dispatch-attached time in the single-stream leg."""
    if entry is None or not alone or not all(alone.get(k) for k in kinds):
        return
    ms = sum(_lib.mean_ms(alone[k]) for k in kinds)
    gbs = entry["algorithmic_bytes"] / (ms * 1e-3) / 1e9
    entry["alone"] = {"avg_ms": ms, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                      "note": "single-stream leg after the timed steps (no weight-gradient stream, task heads on the caller's stream)"}


def lookup_rider_ab(lss, mats, logits_dtype, used_dtype, has_oracle, reps=60):
    """What the calibration lookup costs as a rider of the depth softmax's launch (mmt_depth_softmax_forward_plan_prepare): the
    softmax of the step's shape with and without it, `reps` launches each between two events on the current stream (the lookup's
    own workgroups run beside the softmax's rows; the batch is one the cache knows, as in every timed step).  -> dict of ms."""
    import torch
    from mm_training_amd.ops.bev_geometry import depth_softmax
    B, N = mats["sensor2ego_mats"].shape[0], mats["sensor2ego_mats"].shape[2]
    D, fH, fW = lss.frustum_d.numel(), lss.frustum_v.numel(), lss.frustum_u.numel()
    cache = lss._plan_cache_for(B, N, fH, fW, lss.frustum_d.device, None)
    if cache is None:
        return None
    combine = lss.camera_matrices(mats["sensor2ego_mats"][:, 0], mats["intrin_mats"][:, 0], None)
    lookup = (combine, (lss.frustum_u, lss.frustum_v, lss.frustum_d), lss._voxel_num_host, lss._voxel_coord_host, lss._voxel_size_host, cache)
    g = torch.Generator(device="cuda").manual_seed(5)
    logits = torch.randn(B * N, D, fH, fW, device="cuda", generator=g).to(logits_dtype).contiguous(memory_format=torch.channels_last)
    oracle = None
    if has_oracle:
        oracle = torch.zeros(B * N, D, fH, fW, device="cuda").contiguous(memory_format=torch.channels_last)
        oracle[:, 3, ::2, ::2] = 1.0
    from mm_training_amd import _lib
    out = {}
    with torch.no_grad():
        for name, lk in (("plain", None), ("with_lookup", lookup), ("plain_again", None)):
            for _ in range(5):
                depth_softmax(logits, oracle, used_dtype, plan_lookup=lk)
            saved, _lib.TIMING = _lib.TIMING, {}              # the library's dispatch-attached events around every launch
            try:
                for _ in range(reps):
                    depth_softmax(logits, oracle, used_dtype, plan_lookup=lk)
                torch.cuda.synchronize()
                out[name] = _lib.mean_ms(_lib.TIMING["softmax"])
            finally:
                _lib.TIMING = saved
    plain = min(out["plain"], out["plain_again"])
    return {"softmax_ms": plain, "softmax_with_lookup_ms": out["with_lookup"], "lookup_cost_ms": max(0.0, out["with_lookup"] - plain), "launches_each": reps}


def roofline_entry(kernel, nbytes, ms, traffic=None, l2_bytes=None, note=None):
    gbs = nbytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": nbytes, "avg_ms": ms}
    if l2_bytes is not None:
        l2 = l2_bytes / (ms * 1e-3) / 1e9
        r["l2_side"] = {"bytes": l2_bytes, "achieved": l2, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": l2 / L2_PEAK_GBS}
    if note:
        r["note"] = note
    return r


# ------------------------------------------------------------------------------------------------
# CPU baseline

def _cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, os.cpu_count() or 1, len(phys) or None


def cpu_baseline(geom, feats, vn, grad_out_nhwc, budget_s=25.0):
    """The oracle's torch-CPU port of the reference semantics (scatter_add_ forward, and
    index_add_ as a second line, + masked-gather backward) timed on this host's cores: 3 warm-up calls, then
    >= 10 repetitions (fewer only if the time budget runs out) at the best of a few thread counts (all logical
    cores is pathologically slow for scatter_add_ on a many-core host), median and best, plus a 1-thread figure.
    A reported baseline beside `hotpath_samples_per_s`, not a target."""
    import oracle
    B, P, C = feats.shape
    nx, ny, nz = vn
    t_start = time.perf_counter()

    def one(index_add=False):
        t0 = time.perf_counter()
        out, pos = oracle.torch_forward_scatter_add(geom, feats, nx, ny, nz, use_index_add=index_add)
        t1 = time.perf_counter()
        gi = oracle.torch_backward_gather(pos, grad_out_nhwc)
        t2 = time.perf_counter()
        return t2 - t0, t1 - t0, out, pos, gi

    model, logical, physical = _cpu_info()
    tried = {}
    for th in sorted({min(logical, 16), min(logical, 32), min(logical, 64)}):
        torch.set_num_threads(th)
        if not tried:
            for _ in range(2):
                one()                               # first-touch / allocator warm-up
        tried[th] = one()[0]
        if time.perf_counter() - t_start > 0.3 * budget_s:
            break
    best_threads = min(tried, key=tried.get)
    torch.set_num_threads(best_threads)
    times, fwd_times = [], []
    while len(times) < 3 or (time.perf_counter() - t_start < 0.6 * budget_s and len(times) < 10):
        t, tf, out, pos, gi = one()
        times.append(t)
        fwd_times.append(tf)
    ia = sorted(one(index_add=True)[1] for _ in range(3))
    torch.set_num_threads(1)
    t1s = []
    while len(t1s) < 1 or (time.perf_counter() - t_start < budget_s and len(t1s) < 3):
        t1s.append(one()[0])
    torch.set_num_threads(best_threads)
    times.sort()
    fwd_times.sort()
    med, best = times[len(times) // 2], times[0]
    return {"value": B / med, "unit": "samples/s", "cores": best_threads, "kind": "port",
            "sample": f"{len(times)} reps (after 3 warm-up calls) of voxel_pooling fwd (torch scatter_add_) + bwd (masked gather) on CPU "
                      f"at the full camera shape B={B} P={P} C={C} grid {nx}x{ny}x{nz}: median {med * 1e3:.1f} ms, best {best * 1e3:.1f} ms "
                      f"per fwd+bwd with {best_threads} threads; compare with hotpath_samples_per_s (same op, same shape), "
                      f"NOT with the training-step value",
            "cpu_model": model, "logical_cores": logical, "physical_cores": physical,
            "median_ms": med * 1e3, "best_ms": best * 1e3, "forward_scatter_add_median_ms": fwd_times[len(fwd_times) // 2] * 1e3,
            "forward_index_add_median_ms": ia[len(ia) // 2] * 1e3,
            "threads_tried_ms": {str(k): v * 1e3 for k, v in tried.items()},
            "one_thread": {"value": B / min(t1s), "ms": min(t1s) * 1e3, "reps": len(t1s)}}, out, pos, gi


# ------------------------------------------------------------------------------------------------
# the drop-in voxel_pooling op alone (hotpath mode, and the leg after the training steps)

def kept_count(geom, vn):
    nx, ny, nz = vn
    g3 = geom.reshape(-1, 3)
    kept = ((g3[:, 0] >= 0) & (g3[:, 0] < nx) & (g3[:, 1] >= 0) & (g3[:, 1] < ny) & (g3[:, 2] >= 0) & (g3[:, 2] < nz))
    return int(kept.sum().item())


def hotpath_leg(geom, vn, C, dtype, iters, warmup, seed=100, flags=None):
    """voxel_pooling(geom, feats, vn) forward + backward, `iters` times, every launch carrying dispatch-attached
    events.  Returns (fwd_ms, bwd_ms, feats_cpu, grad_out, out, feats) for the roofline entries and the parity check."""
    from mm_training_amd import _lib, synthetic
    from mm_training_amd.ops import voxel_pooling as vp_pkg
    B = geom.shape[0]
    P = geom[0].numel() // 3
    nx, ny, nz = vn
    feats_cpu = synthetic.features((B, P, C), seed=seed)
    feats = feats_cpu.cuda()
    if dtype == "bf16":
        feats = feats.bfloat16()
    feats.requires_grad_(True)
    grad_out = torch.randn(B, ny, nx, C, generator=torch.Generator().manual_seed(1)).cuda().permute(0, 3, 1, 2)  # channels-last grad
    op = vp_pkg.voxel_pooling_bf16 if dtype == "bf16" else vp_pkg.voxel_pooling

    def step():
        feats.grad = None
        out = op(geom, feats, vn)
        out.backward(grad_out)
        return out

    for _ in range(warmup):
        step()
    saved, _lib.TIMING = _lib.TIMING, {}
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = step()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        timing = _lib.TIMING
    finally:
        _lib.TIMING = saved
    return _lib.mean_ms(timing["forward"]), _lib.mean_ms(timing["backward"]), wall, feats_cpu, grad_out, out, feats


def hotpath_main(args, rank, local_rank, world):
    from mm_training_amd import _lib, synthetic
    _lib.lib()
    shape = HOTPATH_SHAPES[args.config]
    dtype = args.dtype or ("bf16" if args.config == "cfg5" else "f32")
    B, C = shape["batch"], shape["channels"]
    geom_cpu, vn = synthetic.rig_geometry(B, shape["num_cams"], shape["final_dim"], shape["downsample"], shape["d_bound"],
                                          BEV_BOUNDS["x_bound"], BEV_BOUNDS["y_bound"], BEV_BOUNDS["z_bound"], seed=rank)
    nx, ny, nz = vn
    geom = geom_cpu.cuda()
    P = geom_cpu[0].numel() // 3
    barrier(world)
    fwd_ms, bwd_ms, wall, feats_cpu, grad_out, out, feats = hotpath_leg(geom, vn, C, dtype, args.steps, args.warmup, seed=100 + rank)
    barrier(world)
    elapsed = max_over_ranks(wall, world)
    if rank != 0:
        return
    K = kept_count(geom, vn)
    fb = 2 if dtype == "bf16" else 4
    fwd_bytes, bwd_bytes = algorithmic_bytes(B * P, K, C, B, ny, nx, fb)
    fH, fW = shape["final_dim"][0] // shape["downsample"], shape["final_dim"][1] // shape["downsample"]
    cfgname = args.config if dtype == "f32" else f"{args.config}_bf16"
    fwd_kernel = "vp_fwd_seg_gather" + ("_bf16" if dtype == "bf16" else "")
    bwd_kernels = ("vp_bwd_prepare", "vp_bwd_rows_bf16" if dtype == "bf16" else "vp_bwd_rows_vec4")
    res = {
        "metric": "training samples/sec at bs=%d/GPU (hot path only: voxel_pooling fwd+bwd); voxel_pooling HBM GB/s" % B,
        "value": world * B * args.steps / elapsed, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"camera half of {WORKLOADS[args.config].split(':')[0]}: voxel_pooling forward+backward, "
                               f"{shape['num_cams']} cams {shape['final_dim'][0]}x{shape['final_dim'][1]} ds{shape['downsample']} "
                               f"D={int((shape['d_bound'][1] - shape['d_bound'][0]) / shape['d_bound'][2])} fH={fH} fW={fW} C={C}, "
                               f"BEV {nx}x{ny}x{nz}, analytic 6-camera rig geometry, bs={B}/GPU, "
                               f"{'bf16 feature storage, fp32 accumulate, fp32 BEV' if dtype == 'bf16' else 'fp32'}",
                   "global_batch": world * B, "points_per_sample": P, "kept_fraction": K / (B * P),
                   "parallelism": f"dp{world}", "mode": args.mode},
        "roofline": roofline_entry(f"{fwd_kernel} (voxel_pooling forward)", fwd_bytes, fwd_ms, pmc_traffic(cfgname, (fwd_kernel,))),
        "roofline_backward": roofline_entry(f"{' + '.join(bwd_kernels)} (voxel_pooling backward)", bwd_bytes, bwd_ms,
                                            pmc_traffic(cfgname, bwd_kernels)),
        "hotpath_samples_per_s": B / ((fwd_ms + bwd_ms) * 1e-3),
    }
    if world == 1 and not args.no_cpu_baseline:
        # the CPU port sees the same (for bf16: the same ROUNDED) feature values, up-cast to fp32
        f_cpu = feats.detach().float().cpu().reshape(B, P, C)
        base, ref_out, ref_pos, ref_gi = cpu_baseline(geom_cpu.reshape(B, P, 3), f_cpu, vn,
                                                      grad_out.permute(0, 2, 3, 1).contiguous().cpu())
        res["cpu_baseline"] = base
        # same-run parity of the measured path against the CPU port
        err = (out.detach().float().permute(0, 2, 3, 1).cpu() - ref_out).abs().max().item()
        gi = feats.grad.reshape(B, P, C).cpu()
        gi_equal = bool(torch.equal(gi, ref_gi.to(gi.dtype)))     # bf16: exact after rounding the fp32 gather
        res["parity"] = {"bev_max_abs_err": err, "grad_in_bit_exact": gi_equal}
    print(json.dumps(res), flush=True)


# ------------------------------------------------------------------------------------------------
# launcher rehearsal on CPU

def rehearsal_main(args, rank, world):
    """--device cpu: the N>1 plumbing (rank spawn, rendezvous, DDP bucketed all-reduce, barrier, max over ranks,
    rank 0's JSON line) on gloo with the part of the model that is plain PyTorch: the dense detection head on a
    synthetic BEV map.  The HIP hot path has no CPU form and is NOT part of this mode; the line says so."""
    from mm_training_amd.dp.configs import make_config
    from mm_training_amd.layers.heads.bev_depth_head import BEVDepthHead
    cfg = make_config(args.config)
    torch.manual_seed(0)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(world, 1) // 2))
    head = BEVDepthHead(**cfg["head_conf"])
    net = torch.nn.parallel.DistributedDataParallel(head) if world > 1 else head
    opt = torch.optim.AdamW(head.parameters(), lr=1e-4)
    B = cfg["batch_size"]
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(B, cfg["fuse_layer_in_channels"], 128, 128, generator=g)
    boxes = [torch.cat([torch.rand(5, 2, generator=g) * 80 - 40, torch.tensor([[-1.0, 1.9, 4.6, 1.7, 0.3, 0.5, -0.2]]).repeat(5, 1)], 1)
             for _ in range(B)]
    labels = [torch.randint(0, 4, (5,), generator=g) for _ in range(B)]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = head.loss(head.get_targets_torch(boxes, labels), net(x))
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    barrier(world, "cpu")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier(world, "cpu")
    elapsed = max_over_ranks(time.perf_counter() - t0, world, "cpu")
    if rank == 0:
        print(json.dumps({
            "metric": "LAUNCHER REHEARSAL on CPU (not the benchmark): samples/s of the dense detection head", "value": world * B * args.steps / elapsed,
            "unit": "samples/s", "n_gpus": 0, "ranks": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "launcher rehearsal: gloo ranks on CPU, dense detection head only (the HIP hot path needs a GPU)",
                       "global_batch": world * B, "parallelism": f"dp{world}", "mode": "rehearsal-cpu", "final_loss": float(loss.detach())}}), flush=True)


# ------------------------------------------------------------------------------------------------
# training step

def train_main(args, rank, local_rank, world):
    """Full training step of a BASELINE config: synthetic frames -> depth labels -> BEVDepth
    (+LiDAR) forward -> detection + depth loss -> backward (DDP bucketed all-reduce over RCCL,
    overlapped) -> grad clip -> AdamW.  Nothing is skipped inside the timed region."""
    from mm_training_amd import _lib
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    _lib.lib()
    global PMC_RIG
    PMC_RIG = args.rig
    # benchmark=True makes PyTorch ask MIOpen's find API, which is answered from the find DB
    # (only for the configurations the DB was produced on: an unknown shape would start a search)
    db_cfg = SHIPPED_MIOPEN_DB and args.config in ("cfg2", "cfg3", "cfg4", "cfg5", "aim")
    torch.backends.cudnn.benchmark = bool(args.miopen_tune) or db_cfg
    dev = torch.device("cuda", local_rank)
    cfg = make_config(args.config)
    # storage type of the hot-path operands (fp32 accumulate either way); the dense nets keep cfg["dtype"]
    dtype = args.dtype or cfg["hot_path_dtype"]
    cfg["hot_path_dtype"] = dtype
    torch.manual_seed(0)
    import numpy as np
    np.random.seed(0)            # augment_images draws its per-camera flags from numpy's global generator (like the reference)
    if args.aten_softmax:
        os.environ["MMT_ATEN_SOFTMAX"] = "1"
    if args.conv_overlap:
        os.environ["MMT_CONV_OVERLAP"] = args.conv_overlap
    if args.head_streams is not None:
        os.environ["MMT_HEAD_STREAMS"] = str(args.head_streams)
    if args.no_augment:
        cfg["augment_images"] = False
    if args.no_depth_oracle:
        cfg["use_depth_loss"] = False
    ts = TrainStep(cfg, dev, world_size=world)
    if args.full_lidar_canvas and cfg["use_lidar"]:
        ts.model.full_lidar_canvas = True
    fused = False
    if cfg["use_cam"]:
        if args.unfused or args.cached_plan:
            ts.model.backbone.fused_lift_splat = False
        fused = bool(ts.model.backbone.fused_lift_splat)
        ts.model.backbone.lift_splat_backward = args.lift_splat_backward
        if args.geom_form:
            ts.model.backbone.camera_form = False
    B = cfg["batch_size"]
    # a small pool of distinct pre-generated batches resident in HBM (input is never the bottleneck)
    rig = args.rig
    if rig == "nuscenes":                    # the reference fixture's calibration (data: tests/golden/nusc_rig.npz, made by tests/golden/make_golden.py)
        import numpy as np
        g_ = np.load(os.path.join(ROOT, "tests", "golden", "nusc_rig.npz"))
        rig = (g_["sensor2ego"], g_["intrin"], tuple(int(v) for v in g_["image_hw"]))
    batches = [synthetic_batch(cfg, dev, seed=1000 * rank + i, rig=rig) for i in range(2)]
    if args.cached_plan or args.calibration_ids:
        # each synthetic batch has its own (jittered) rig; its id lets LSSFPN reuse what depends on the
        # calibration alone (unfused path: the sort of the points by BEV cell; fused path: the camera matrices, the
        # geometry's column summary and the choice of the backward kernel)
        for i, b in enumerate(batches):
            b[1]["calibration_id"] = ("synthetic", 1000 * rank + i)
    for i in range(args.warmup):
        ts(batches[i % len(batches)])
    if os.environ.get("MMT_BENCH_FAIL_RANK") == str(rank):
        # test hook (tests/test_bench_ranks_gpu.py): this rank dies after its warm-up; the launcher must report it and take
        # the ranks waiting in the barrier down instead of hanging
        print(f"[bench] rank {rank}: MMT_BENCH_FAIL_RANK set, exiting with code 3", file=sys.stderr, flush=True)
        os._exit(3)
    _lib.TIMING = {}
    barrier(world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, det, dep = ts(batches[i % len(batches)])
    barrier(world)
    elapsed = time.perf_counter() - t0
    timing, _lib.TIMING = _lib.TIMING, None
    elapsed = max_over_ranks(elapsed, world)
    alone = None
    if world == 1 and (ts.conv_overlap in ("deferred", "pair") or ts.model.head.task_streams > 1):
        alone = single_stream_leg(ts, batches, 6, _lib)
    dinfo = distributed_info(world, local_rank, ts)          # (gathers: every rank)
    if rank != 0:
        barrier(world)          # rank 0 adds its roofline legs below; everybody leaves together
        return
    from mm_training_amd import miopen_db
    db_status = miopen_db.status()          # did MIOpen read the shipped find DB (same build string), or write files of its own?
    res = {
        "metric": "training samples/sec at bs=%d/GPU; voxel_pooling HBM GB/s" % B,
        "value": world * B * args.steps / elapsed, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": WORKLOADS[args.config],
                   "global_batch": world * B, "parallelism": f"dp{world}", "mode": "train",
                   "params_M": sum(p.numel() for p in ts.model.parameters()) / 1e6,
                   "final_loss": float(loss), "miopen_exhaustive_search": bool(args.miopen_tune),
                   "miopen_shipped_find_db": bool(db_cfg) and bool(db_status.get("matched")), "miopen_find_db": db_status, "fused_lift_splat": fused, "cached_plan": bool(args.cached_plan),
                   "dense_nets_dtype": "bf16 autocast" if ts.amp_dtype is not None else "f32",
                   "hot_path_storage_dtype": dtype,
                   # exps/mm_training_aim.py:258-259: both run inside every timed step
                   "rig": args.rig,
                   "augment_images": bool(ts.augment and cfg["use_cam"]), "depth_oracle": bool(ts.pass_depth_labels and cfg["use_cam"]),
                   "conv_weight_gradients": {None: "same stream (autograd)", "inline": "same stream"}.get(ts.conv_overlap, ts.conv_overlap),
                   "task_head_streams": int(ts.model.head.task_streams),
                   # the 24 CenterPoint branches: first ConvModules as one wide layer, final convolutions in csrc/thin_conv.hip
                   # ("auto": when applicable -- training on the GPU; MMT_HEAD_FUSED / MMT_HEAD_FINALS switch them off)
                   "task_heads_fused": {"first_layer": ts.model.head.fuse_branch_stems, "final_convolutions": bool(ts.model.head.fuse_final_convs)},
                   # exps/conf_aim.py:57: the image backbone's stem has no gradients, its BatchNorm runs in eval mode
                   "image_backbone_frozen_stages": (int(ts.model.backbone.img_backbone.frozen_stages) if cfg["use_cam"] else None),
                   "distributed": dinfo},
    }
    fb = 2 if dtype == "bf16" else 4
    geom = vn = None
    if cfg["use_cam"]:
        lss = ts.model.backbone
        with torch.no_grad():
            m = batches[0][1]
            geom = lss.get_geometry_voxels(m["sensor2ego_mats"][:, 0], m["intrin_mats"][:, 0])
        vn = list(lss._voxel_num_host)
        nx, ny, nz = vn
        K, BP, C = kept_count(geom, vn), geom.numel() // 3, lss.output_channels
        BN_HW = BP // lss.depth_channels
        res["config"]["kept_fraction"] = K / BP
        if fused and timing.get("lift_splat_forward"):
            fwd_ms, bwd_ms = _lib.mean_ms(timing["lift_splat_forward"]), _lib.mean_ms(timing["lift_splat_backward"])
            from mm_training_amd.ops.bev_geometry import last_kernel_family
            fam_f, fam_b = last_kernel_family(), last_kernel_family(backward=True)
            fam_f_detail = last_kernel_family(detail=True)
            camera = fam_f.endswith("+camera")
            fbytes, bbytes, l2f, l2b = lift_splat_bytes(BP, K, C, B, BN_HW, ny, nx, fb, camera_form=camera)
            note = ("fused get_geometry + quantise + lift + voxel_pooling (SURVEY 8 rows f1 + f3): neither the [B*P, C] feature matrix nor "
                    "(camera form) the geom tensor exists -- the kernels compute every point's cell from the B*N camera matrices with the "
                    "arithmetic of mmt_frustum_geometry, so the HBM-side algorithmic bytes are ~13x below the drop-in op's (that also lowers "
                    "this fraction: fewer bytes for the same atomic-bound time).  The forward is bound by the memory-side fp32 atomic units "
                    "(about 1.25 TB/s of added bytes in whole 64-byte segments, tools/ubench/atomic_rows.hip: ~25 MB of BEV rows per launch at "
                    "this shape = `atomic_side`); its zero-fill of the BEV map is part of the timed sequence (MMT_LSS_ZERO_OUTPUT).  The "
                    "backward (matrix-core column kernel on a level rig, else a per-pixel ray walk) by its latency-bound load phase followed by "
                    "tile staging, neither by sustained HBM bandwidth -- l2_side prices the rows a point-wise gather moves against the "
                    "aggregate L2 bandwidth; the drop-in op's HBM roofline is roofline_voxel_pooling[_backward]")
            sfx = "_bf16" if dtype == "bf16" else ""
            tiles = fam_f.startswith("tile")
            plan = fam_f.startswith("plan")
            kfwd = {"ray": "lss_ray_fwd_reg" if "+register" in fam_f_detail else ("lss_ray_fwd_blk" if "+block" in fam_f_detail else "lss_ray_fwd"),
                    "tile": "lss_splat_fwd_tile", "plan": "lss_plan_fwd"}.get(fam_f.split("+")[0], fam_f)
            if plan:
                note = ("fused get_geometry + quantise + lift + voxel_pooling (SURVEY 8 rows f1 + f3) in its PLAN form: output-stationary on a "
                        "per-calibration plan learnt on the device (cell -> runs of (column, row block, bins)); a workgroup takes a job (cells of an "
                        "8 x 8 BEV tile, <= 96 runs), sums depth * context per run in registers, one partial row per run into LDS, then sums every "
                        "cell's partial rows in plan order and STORES -- no zero fill, no atomics, bit-identical from step to step.  `avg_ms` is the "
                        "forward CHAIN: the kernel + the per-step lookup of the batch's calibrations (no launch of its own: it rides in the depth softmax's "
                        "launch, and there is none at all while named calibrations repeat) -- `parts` has the split.  The kernel is bound by the context rows it re-reads through "
                        "L1 (a column's 16 rows once per job it crosses: ~90 MB per launch at BASELINE configs[3]), not by HBM: l2_side")
            kbwd = {"ray": "lss_ray_bwd", "tile": "lss_splat_bwd_tile", "column": "lss_col_bwd"}.get(fam_b.split("+")[0], fam_b)
            column = fam_b.startswith("column")
            adaptive = lss._column_adaptive
            res["config"]["lift_splat_kernels"] = {
                "forward": fam_f_detail, "backward": fam_b,
                "plan_form": ("on: output-stationary forward on the learnt plan; calls / hits / calibrations learnt / samples served by the "
                              "brute-force path: %s" % lss.plan_cache_counters()) if plan else "off",
                "exclusive_cell_cache": ("on: runs into single-run cells are stored, not added (learnt on the device per calibration; "
                                         "the synthetic batch repeats, so every timed step uses it)") if "+exclusive" in fam_f_detail else "off",
                "exclusive_cell_cache_calls": lss.exclusive_cache_counters() if "+exclusive" in fam_f_detail else None,
                "geometry": "computed in the kernels (camera form)" if camera else "geom tensor (mmt_frustum_geometry every step)",
                "backward_choice": lss.lift_splat_backward if lss.lift_splat_backward != "auto" else (
                    "auto: per calibration id, from the geometry" if args.calibration_ids else
                    "auto: from the column kernel's own counters, read back lazily (share of kept points outside their column's cell: %s)"
                    % (None if adaptive is None else adaptive["share"]))}
            key_f = ("lift_splat_forward_plan" if plan else "lift_splat_forward_tile" if tiles else ("lift_splat_forward_camera" if camera else "lift_splat_forward")) + sfx
            key_b = ("lift_splat_backward_tile" if tiles else (("lift_splat_backward_column" if column else "lift_splat_backward") + ("_camera" if camera else ""))) + sfx
            # the timed sequence of the forward is the zero fill + the kernel: so is its traffic
            if plan:
                # "sum of all kernels launched by one op call" -- the plan form's per-step lookup of the batch's
                # calibrations (mmt_lss_plan_prepare) belongs to the forward's chain: avg_ms / achieved / frac are the CHAIN's;
                # `parts` keeps the split.  With mats_dict['calibration_id'] the
                # module skips the lookup while the ids repeat (0 launches per step in steady state).
                prep = timing.get("lift_splat_plan_prepare") or []
                prep_ms = sum(s_.elapsed_time(e_) for s_, e_ in prep) / max(len(timing["lift_splat_forward"]), 1)
                # without ids the lookup RIDES in the depth softmax's launch (mmt_depth_softmax_forward_plan_prepare: its 1 + min(B, 8)
                # workgroups in front of the softmax's grid): no launch of its own; what it adds to that launch is measured here, A/B
                rider = None
                if not prep and not args.calibration_ids and os.environ.get("MMT_PLAN_LOOKUP_RIDER", "1") != "0":
                    rider = lookup_rider_ab(lss, batches[0][1], torch.bfloat16 if ts.amp_dtype is not None else torch.float32,
                                            torch.bfloat16 if dtype == "bf16" else torch.float32, bool(ts.pass_depth_labels))
                ride_ms = rider["lookup_cost_ms"] if rider else 0.0
                res["roofline"] = roofline_entry(f"{kfwd}{sfx} + the per-step calibration lookup (fused lift-splat forward, plan form = the step's "
                                                 "voxel_pooling forward)", fbytes, fwd_ms + prep_ms + ride_ms, pmc_traffic(args.config, (key_f,)), l2f, note)
                res["roofline"]["parts"] = {"forward_kernel_ms": fwd_ms, "lookup_ms_per_step": prep_ms + ride_ms,
                                            "lookup_launches_per_step": 1.0 * len(prep) / max(len(timing["lift_splat_forward"]), 1),
                                            "lookup_kernels": ("none: the named calibrations repeat, the verdicts in their cache stand" if args.calibration_ids and not prep else
                                                               "rides in the depth softmax's launch (lss_plan_lookup_softmax: the lookup's workgroups in front of the softmax's grid; "
                                                               "a batch seen before is recognised from a snapshot); lookup_ms_per_step = that launch with the lookup minus without, `lookup_rider_ab`"
                                                               if rider else "lss_plan_lookup (probe + build in one launch; a batch seen before is recognised from a snapshot)"),
                                            "lookup_rider_ab": rider,
                                            "forward_kernel_frac": fbytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            else:
                res["roofline"] = roofline_entry(f"lss_zero_fill + {kfwd}{sfx} (fused lift-splat forward = the step's voxel_pooling forward)", fbytes, fwd_ms,
                                                 pmc_traffic(args.config, (key_f, "lss_zero_fill")), l2f, note)
            atomic_bytes = None
            try:
                newest = [r for r in ("r05", "r04", "r03") if os.path.exists(os.path.join(ROOT, "profiles", f"{r}_pmc_{args.config}.json"))][0]
                atomic_bytes = float(json.load(open(os.path.join(ROOT, "profiles", f"{newest}_pmc_{args.config}.json")))["kernels"][key_f]["atomic_bytes"])
            except Exception:
                pass
            if atomic_bytes and not plan:
                ab = atomic_bytes / (fwd_ms * 1e-3) / 1e9
                res["roofline"]["atomic_side"] = {"bytes": atomic_bytes, "achieved": ab, "peak": ATOMIC_PEAK_GBS, "unit": "GB/s", "frac": ab / ATOMIC_PEAK_GBS,
                                                  "note": "TCC_EA0_ATOMIC x 64 B per launch (PMC pass) against the chip-wide memory-side fp32 atomic rate"}
            res["roofline_backward"] = roofline_entry(f"{kbwd}{sfx} (fused lift-splat backward = the step's voxel_pooling backward)", bbytes, bwd_ms,
                                                      pmc_traffic(args.config, (key_b,)), l2b)
        elif timing.get("forward"):
            fwd_ms, bwd_ms = _lib.mean_ms(timing["forward"]), _lib.mean_ms(timing["backward"])
            fbytes, bbytes = algorithmic_bytes(BP, K, C, B, ny, nx, fb)
            if args.cached_plan and lss._plan_cache:
                plan = next(iter(lss._plan_cache.values()))
                # cached sort: no geom read / pos_memo write per step; row ids + item descriptors instead
                fbytes = 4 * C * K + 4 * K + 16 * plan.num_items + 4 * C * B * ny * nx
                res["roofline"] = roofline_entry("vp_planned_items + vp_planned_fold (cached-plan voxel_pooling forward, inside the training step)",
                                                 fbytes, fwd_ms)
            else:
                res["roofline"] = roofline_entry("vp_fwd_seg_gather (voxel_pooling forward, inside the training step)", fbytes, fwd_ms,
                                                 pmc_traffic(args.config + "_unfused", ("vp_fwd_seg_gather",)))
            res["roofline_backward"] = roofline_entry("vp_bwd_prepare + vp_bwd_rows_vec4 (voxel_pooling backward, inside the training step)",
                                                      bbytes, bwd_ms, pmc_traffic(args.config + "_unfused", ("vp_bwd_prepare", "vp_bwd_rows_vec4")))
    if cfg["use_cam"] and timing.get("softmax"):
        # lss_fpn.py:423 (+ :427-438): logits read once, probabilities written once (+ the oracle rows read and the depth the
        # lift uses written when the labels are passed in / the operand is bf16); backward: probabilities + the two consumers'
        # gradients (+ the oracle rows for the foreground test) read, the logit gradient written
        pix_d = BP                                     # pixels * D
        lb = 2 if ts.amp_dtype is not None else 4      # bytes per logit (bf16 under autocast)
        has_oracle, used_b = bool(ts.pass_depth_labels), fb
        separate_used = has_oracle or dtype == "bf16"
        sm_f = lb * pix_d + 4 * pix_d + (4 * pix_d if has_oracle else 0) + (used_b * pix_d if separate_used else 0)
        sm_b = 4 * pix_d + 4 * pix_d + (used_b * pix_d if separate_used else 0) + (4 * pix_d if has_oracle else 0) + lb * pix_d
        t_f, t_b = _lib.mean_ms(timing["softmax"]), _lib.mean_ms(timing["softmax_backward"])
        res["roofline_softmax"] = roofline_entry("depth_softmax_fwd (depth distribution + oracle overwrite, pixel-major; lss_fpn.py:423-438)", sm_f, t_f,
                                                 pmc_traffic(args.config, ("depth_softmax_fwd",)))
        res["roofline_softmax"]["backward"] = roofline_entry("depth_softmax_bwd", sm_b, t_b, pmc_traffic(args.config, ("depth_softmax_bwd",)))
    if cfg["use_cam"] and timing.get("dcn_forward"):
        # DepthNet's deformable convolution (lss_fpn.py:189-197) as implicit GEMMs: bound by the fp32 matrix rate
        # (exact-fp32 MFMA), the HBM side beside it.  One GEMM forward, two backward (weight + data gradient).
        dcn = [m for m in ts.model.modules() if type(m).__name__ == "DeformConv2dPack"][0]
        dB, dC, dH, dW, dO, dG = dcn.last_shape
        flop = 2.0 * dB * dH * dW * 9 * (dC // dG) * dO
        io_f = 4.0 * (dB * dH * dW * (dC + 18 + dO) + dcn.weight.numel())
        io_b = 4.0 * (dB * dH * dW * (2 * dC + 2 * 18 + dO) + 2 * dcn.weight.numel())
        t_f = _lib.mean_ms(timing["dcn_forward"])
        t_b = _lib.mean_ms(timing["dcn_backward"]) if timing.get("dcn_backward") else None

        def dcn_entry(kernels, nflop, nbytes, ms):
            tf = nflop / (ms * 1e-3) / 1e12
            e = {"bound": "mfma", "kernel": kernels, "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                 "flop": nflop, "avg_ms": ms, "algorithmic_bytes": nbytes,
                 "hbm_side": {"achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
            return e
        fwd_k = ("dcn_pack_weights", "dcn_fwd_mfma")
        bwd_k = ("dcn_pack_weights", "dcn_plan_taps", "dcn_wgrad_mfma", "dcn_wgrad_reduce", "dcn_dgrad_gather", "dcn_offset_reduce_parts")
        r = dcn_entry(" + ".join(fwd_k) + " (deformable 3x3 convolution forward as one implicit GEMM: taps sampled into LDS, no column buffer)",
                      flop, io_f, t_f)
        r["traffic"] = pmc_traffic(args.config, fwd_k)
        r["shape"] = {"x": [dB, dC, dH, dW], "out_channels": dO, "groups": dG, "operands": "f32", "accumulate": "f32 (v_mfma_f32_32x32x2_f32)"}
        if t_b is not None:
            r["backward"] = dcn_entry(" + ".join(bwd_k) + " (weight gradient + data / offset gradient: two implicit GEMMs, the scatter of the data "
                                      "gradient as a gather through per-destination lists)", 2 * flop, io_b, t_b)
            r["backward"]["traffic"] = pmc_traffic(args.config, bwd_k)
        else:
            r["backward"] = {"kernel": "column form: dcn_im2col (columns rebuilt in the backward, not stored by the forward) + vendor GEMMs + dcn_plan + "
                                       "dcn_col2im_gather + dcn_offset_reduce", "note": f"frame of {dH * dW} pixels: above the 768 of the gather form of the "
                                       "implicit data gradient (mmt_dcn_backward_form == 1); not timed as one sequence (the GEMMs are the vendor library's)"}
        r["note"] = ("x + offsets + weights + out (and their gradients) are the algorithmic bytes: the [B*H*W, 9*C] fp32 columns of the im2col form "
                     "(311 MB at this shape, written once and read by three GEMM passes) no longer exist")
        res["roofline_dcn"] = r
    if cfg["use_cam"] and timing.get("bev_warp"):
        # BEV augmentation warp of the pooled camera map into the camera|LiDAR buffer (models/bev_depth.py:69-84):
        # forward = every output cell reads 4 input rows (a row is read by ~4 cells: L2 hits) and writes one; backward = the same gather transposed
        nxw, nyw = vn[0], vn[1]
        wb = 4.0 * B * nyw * nxw * ts.model.backbone.output_channels
        res["roofline_bev_warp"] = roofline_entry("bev_warp_kernel (bilinear affine warp, channels-last, into the concat buffer)", 2 * wb,
                                                  _lib.mean_ms(timing["bev_warp"]), pmc_traffic(args.config, ("bev_warp_kernel",)))
        if timing.get("bev_warp_backward"):
            res["roofline_bev_warp"]["backward"] = roofline_entry("bev_warp_backward_gather (the warp's adjoint as a gather: bit-reproducible)", 2 * wb,
                                                                  _lib.mean_ms(timing["bev_warp_backward"]), pmc_traffic(args.config, ("bev_warp_backward_gather",)))
    if cfg["use_lidar"] and timing.get("voxelize") and timing.get("scatter"):
        enc = ts.model.lidar_encoder
        from mm_training_amd.lidar import hard_voxelize_mean_batch
        with torch.no_grad():
            _, _, co, cnt, _ = hard_voxelize_mean_batch([p.float() for p in batches[0][2]], enc.voxel_size, enc.point_cloud_range,
                                                         enc.max_num_points, enc.max_voxels, enc.num_features, materialize_voxels=False)
        M = int(cnt.sum().item())
        total_pts = sum(int(p.shape[0]) for p in batches[0][2])
        ny_l, nx_l = enc.output_shape
        sampled = None
        if cfg["use_cam"] and not ts.model.full_lidar_canvas and vn is not None and ny_l % vn[1] == 0 and nx_l % vn[0] == 0:
            sy, sx = ny_l // vn[1], nx_l // vn[0]
            m_s = int(((co[:, 0] >= 0) & (co[:, 2] % sy == 0) & (co[:, 3] % sx == 0)).sum().item())
            sampled = (m_s, vn[1], vn[0])
        vox_b, scat_b, scat_bwd_b = lidar_bytes(cfg["point_features"], total_pts, M, enc.num_features, enc.in_channels, B, ny_l, nx_l,
                                                sampled=sampled)
        t_vox, t_scat = _lib.mean_ms(timing["voxelize"]), _lib.mean_ms(timing["scatter"])
        k_scat = "scatter_write_strided_table_kernel" if sampled else "scatter_write_nhwc_table_kernel"
        r = roofline_entry("vox_cells + vox_own + vox_emit (voxelize + mean, region-owner form) and " +
                           ("scatter_write_strided_table (only the pillar cells the nearest resize samples, straight from the voxelizer's table "
                            "into the camera|LiDAR buffer)" if sampled else
                            "scatter_write_nhwc_table (pillar scatter straight from the voxelizer's table)") + ", inside the training step",
                           vox_b + scat_b, t_vox + t_scat, pmc_traffic(args.config, ("vox_cells", "vox_own", "vox_emit", k_scat)))
        if sampled:
            r["sampled_scatter"] = {"stride": [ny_l // vn[1], nx_l // vn[0]], "voxels_on_sampled_cells": sampled[0], "output": [B, sampled[1], sampled[2], enc.in_channels]}
        r["parts"] = {"voxelize_mean": {"algorithmic_bytes": vox_b, "avg_ms": t_vox, "GBps": vox_b / t_vox / 1e6,
                                        "note": "three launch- and latency-bound kernels (cells / own / emit: ~2.5 us of dispatch each + 2-4 dependent round trips), no memory atomics"},
                      "pillar_scatter": {"algorithmic_bytes": scat_b, "avg_ms": t_scat, "GBps": scat_b / t_scat / 1e6}}
        r["voxels"] = M
        res["roofline_lidar"] = r
        if timing.get("scatter_backward"):
            k_sb = "scatter_backward_strided_kernel" if sampled else "scatter_backward_nhwc_unique_kernel"
            res["roofline_lidar_backward"] = roofline_entry(f"{k_sb} (pillar scatter backward)", scat_bwd_b,
                                                            _lib.mean_ms(timing["scatter_backward"]), pmc_traffic(args.config, (k_sb,)))
    if cfg["use_cam"] and not args.no_hotpath_leg and (world == 1 or args.hotpath_leg):
        # the drop-in op at the same shape and geometry, right after the timed steps: the like-for-like figure
        # beside cpu_baseline and the BASELINE metric's "voxel_pooling HBM GB/s"
        gsh = geom.reshape(B, -1, 3).contiguous()
        P = gsh.shape[1]
        fwd_ms, bwd_ms, _, feats_cpu, grad_out, out, feats = hotpath_leg(gsh, vn, C, dtype, 20, 3)
        fbytes, bbytes = algorithmic_bytes(BP, K, C, B, ny, nx, fb)
        sfx = "_bf16" if dtype == "bf16" else ""
        tag = (args.config if args.config != "cfg4" else "cfg2") + sfx      # cfg4's camera half IS the cfg2 shape
        res["roofline_voxel_pooling"] = roofline_entry(f"vp_fwd_seg_gather{sfx} (drop-in voxel_pooling forward, same shape and geometry, "
                                                       "timed after the steps)", fbytes, fwd_ms, pmc_traffic(tag, ("vp_fwd_seg_gather" + sfx,)))
        bk = ("vp_bwd_prepare", "vp_bwd_rows_bf16" if dtype == "bf16" else "vp_bwd_rows_vec4")
        res["roofline_voxel_pooling_backward"] = roofline_entry(f"{' + '.join(bk)} (drop-in voxel_pooling backward)", bbytes, bwd_ms,
                                                                pmc_traffic(tag, bk))
        res["hotpath_samples_per_s"] = B / ((fwd_ms + bwd_ms) * 1e-3)
        if world == 1 and not args.no_cpu_baseline:
            f_cpu = feats.detach().float().cpu().reshape(B, P, C)
            base, ref_out, ref_pos, ref_gi = cpu_baseline(gsh.cpu(), f_cpu, vn, grad_out.permute(0, 2, 3, 1).contiguous().cpu())
            res["cpu_baseline"] = base
            err = (out.detach().float().permute(0, 2, 3, 1).cpu() - ref_out).abs().max().item()
            gi = feats.grad.reshape(B, P, C).cpu()
            res["parity"] = {"bev_max_abs_err": err, "grad_in_bit_exact": bool(torch.equal(gi, ref_gi.to(gi.dtype)))}
    if alone:
        fwd_kind, bwd_kind = ("lift_splat_forward", "lift_splat_backward") if alone.get("lift_splat_forward") else ("forward", "backward")
        add_alone(res.get("roofline"), alone, (fwd_kind,), _lib)
        add_alone(res.get("roofline_backward"), alone, (bwd_kind,), _lib)
        add_alone(res.get("roofline_softmax"), alone, ("softmax",), _lib)
        add_alone((res.get("roofline_softmax") or {}).get("backward"), alone, ("softmax_backward",), _lib)
        add_alone(res.get("roofline_lidar"), alone, ("voxelize", "scatter"), _lib)
        add_alone(res.get("roofline_lidar_backward"), alone, ("scatter_backward",), _lib)
        # (the warp's backward is the first hand-written kernel of the backward pass: in the multi-stream step its dispatch-attached
        # time is almost all time shared with the weight-gradient stream -- ~0.2 ms for a 12 us kernel)
        add_alone(res.get("roofline_bev_warp"), alone, ("bev_warp",), _lib)
        add_alone((res.get("roofline_bev_warp") or {}).get("backward"), alone, ("bev_warp_backward",), _lib)
        res["config"]["streams_note"] = (
            "the timed steps run on several HIP streams (weight gradients of the convolutions on a low-priority side stream, task heads on "
            "two): `avg_ms` / `frac` of a roofline object are the kernel's dispatch-attached time INSIDE those steps, i.e. while it may "
            "share the card with queued weight-gradient kernels (the pooling backward does); `alone` = the same measurement in 6 further "
            "single-stream steps after the timed region (not part of `value`), which is what rocprofv3's per-kernel duration shows")
    print(json.dumps(res), flush=True)
    barrier(world)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, argv))
    _late_imports(args.device)
    rank, local_rank, world = init_dist(args.gpus, args.device, args.shared_gpu_rehearsal)
    try:
        if args.device == "cpu":
            rehearsal_main(args, rank, world)
        elif args.mode == "train":
            train_main(args, rank, local_rank, world)
        else:
            hotpath_main(args, rank, local_rank, world)
    finally:
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
