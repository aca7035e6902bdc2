"""GPU: the image augmentation of the training step against arrays the reference's OWN
augment_images (exps/mm_training_aim.py:88-112) and normalize_images (:510-512) produced in the build container
(tests/golden/make_golden.py::make_augment_images -> augment_images.npz), and the training step's use of it.

  * augment_images(images, labels, 'train') with numpy's global generator seeded like the reference run: the SAME flags,
    images and label maps bit for bit (a flip moves values, it does not compute); 'val' hands its inputs back;
  * normalize_flip_images (normalise + flip in one pass, the step's form): bit-identical to the torch expression on the GPU,
    within 1e-6 absolute of the reference's CPU result (ATen divides by a scalar with a true division on the
    CPU and with a multiplication by the fp32 reciprocal on the GPU -- the reference trains on the GPU);
  * depth_labels(..., flipped) == hflip(depth_labels(...)): the label half of augment_images folded into the label write;
  * TrainStep.forward_loss: draws the flags, hands them to the model as mats['flipped'] and the labels as the depth oracle
    (:258-259), and equals the unfused composition of the reference's steps on the same flags."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_augment_images_matches_the_reference(mmt_lib, golden):
    from mm_training_amd.dp.trainer import augment_images
    g = golden["augment_images"]
    norm = torch.from_numpy(g["normalized"]).cuda()
    labels = torch.from_numpy(g["labels"]).cuda()
    np.random.seed(int(g["seed"]))
    imgs, labs, flips = augment_images(norm, labels, 'train')
    assert isinstance(flips, np.ndarray) and flips.dtype == bool and np.array_equal(flips, g["flips"])
    assert np.array_equal(imgs.cpu().numpy(), g["aug_images"])
    assert np.array_equal(labs.cpu().numpy(), g["aug_labels"])
    assert imgs.shape == norm.shape and labs.shape == labels.shape
    # the flagged cameras really moved, the others did not
    for i, f in enumerate(g["flips"]):
        same = torch.equal(imgs.reshape(-1, *imgs.shape[3:])[i], norm.reshape(-1, *norm.shape[3:])[i])
        assert same == (not f)
    iv, lv, fv = augment_images(norm, labels, 'val')
    assert iv is norm and lv is labels and fv.dtype == bool and not fv.any() and fv.shape == flips.shape


@pytest.mark.parametrize("channels_last", [True, False])
def test_normalize_flip_images(mmt_lib, golden, channels_last):
    from mm_training_amd.dp.trainer import IMG_MEAN, IMG_STD
    from mm_training_amd.ops.train_targets import camera_flags_to_device, normalize_flip_images
    g = golden["augment_images"]
    raw = torch.from_numpy(g["raw"]).cuda()
    fl = camera_flags_to_device(g["flips"], raw.device)
    out = normalize_flip_images(raw, IMG_MEAN, IMG_STD, fl, channels_last=channels_last)
    B, S, N, _, H, W = raw.shape
    assert out.shape == (B, S, N, 3, H, W)
    flat = out.reshape(B * S * N, 3, H, W)
    assert flat.is_contiguous(memory_format=torch.channels_last) == channels_last and flat.data_ptr() == out.data_ptr()
    ref = g["aug_images"]
    err = np.abs(out.cpu().numpy() - ref)
    assert err.max() <= 1e-6          # one ulp of x / 255 (6e-8), divided by std ~ 0.225
    # the torch expression on the GPU (what the reference's lines evaluate to there), flipped per camera: bit for bit
    mean = torch.tensor(IMG_MEAN, device="cuda").view(1, 1, 1, 3, 1, 1)
    std = torch.tensor(IMG_STD, device="cuda").view(1, 1, 1, 3, 1, 1)
    t = (raw[:, :, :, :3] / 255.0 - mean) / std
    t = torch.where(torch.from_numpy(g["flips"]).cuda().view(B, S, N, 1, 1, 1), t.flip(-1), t)
    assert torch.equal(out, t)
    # another width (a four-pixels-per-thread variant of the kernel was measured slower -- 31.9 against 26.6 us at configs[3] -- and
    # dropped: its 16-byte stores land 48 bytes apart; one pixel per thread writes whole 768-byte runs per wave)
    narrow = raw[..., :46].contiguous()
    tn = (narrow[:, :, :, :3] / 255.0 - mean) / std
    tn = torch.where(torch.from_numpy(g["flips"]).cuda().view(B, S, N, 1, 1, 1), tn.flip(-1), tn)
    assert torch.equal(normalize_flip_images(narrow, IMG_MEAN, IMG_STD, fl, channels_last=channels_last), tn)
    # no flags: plain normalize_images
    plain = normalize_flip_images(raw, IMG_MEAN, IMG_STD, None, channels_last=channels_last)
    assert torch.equal(plain, (raw[:, :, :, :3] / 255.0 - mean) / std)
    assert np.abs(plain.cpu().numpy() - g["normalized"]).max() <= 1e-6


def test_hflip_and_label_flip_inside_the_label_kernel(mmt_lib):
    from mm_training_amd.dp import make_config, synthetic_batch
    from mm_training_amd.ops.train_targets import camera_flags_to_device, depth_labels, hflip
    cfg = make_config("tiny")
    dev = torch.device("cuda", 0)
    imgs, mats, pcs, _, _ = synthetic_batch(cfg, dev, seed=11)
    B, S, N, _, H, W = imgs.shape
    ds, db = cfg["backbone_conf"]["downsample_factor"], cfg["backbone_conf"]["d_bound"]
    D = len(torch.arange(*db))
    flips = np.array([True, False, True, True])[:B * N]
    fl = camera_flags_to_device(flips, dev)
    args = (pcs, mats["extrinsics"][:, 0], mats["intrin_mats"][:, 0], mats["bda_mat"], (H, W), ds, db, D)
    plain, plain_bins = depth_labels(*args, return_bins=True)
    flipped, flipped_bins = depth_labels(*args, return_bins=True, flipped=fl)
    fH, fW = H // ds, W // ds
    want = hflip(plain.view(B * N, fH, fW, D), fl)
    assert torch.equal(flipped.view(B * N, fH, fW, D), want)
    ref = torch.where(torch.from_numpy(flips).cuda().view(-1, 1, 1, 1), plain.view(B * N, fH, fW, D).flip(2), plain.view(B * N, fH, fW, D))
    assert torch.equal(want, ref) and not torch.equal(want, plain.view(B * N, fH, fW, D))
    assert torch.equal(flipped_bins.view(B * N, fH, fW), torch.where(torch.from_numpy(flips).cuda().view(-1, 1, 1),
                                                                      plain_bins.view(B * N, fH, fW).flip(2), plain_bins.view(B * N, fH, fW)))
    # images: NCHW planes flipped per camera (group = channels)
    x = torch.randn(B * N * 3, 8, 12, 1, device="cuda")
    y = hflip(x, fl, group=3)
    assert torch.equal(y.view(B * N, 3, 8, 12), torch.where(torch.from_numpy(flips).cuda().view(-1, 1, 1, 1), x.view(B * N, 3, 8, 12).flip(-1),
                                                           x.view(B * N, 3, 8, 12)))
    with pytest.raises(RuntimeError):
        hflip(x, fl, group=2)


def test_training_step_takes_both_branches(mmt_lib):
    """exps/mm_training_aim.py:256-268: labels -> normalise -> augment -> mats['flipped'] -> depth oracle -> model.  The step's
    fused form equals the composition of the reference's functions on the same flags, the model receives the flags and the
    oracle, and switching either branch off changes the loss."""
    from mm_training_amd.dp import TrainStep, make_config, synthetic_batch
    from mm_training_amd.dp import trainer as T
    cfg = make_config("tiny")
    assert cfg["augment_images"] and cfg["use_depth_loss"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ts = TrainStep(cfg, dev)
    assert ts.augment and ts.pass_depth_labels
    ts.model.eval()
    batch = synthetic_batch(cfg, dev, seed=4)
    seen = {}
    real_forward = ts.model.forward

    def spy(x, mats_dict, lidar_oracle=None, timestamps=None):
        seen.update(images=x[0], flipped=mats_dict["flipped"], oracle=lidar_oracle)
        return real_forward(x, mats_dict, lidar_oracle, timestamps)

    ts.model.forward = spy
    ts.net = ts.model
    np.random.seed(123)
    with torch.no_grad():
        loss, det, dep = ts.forward_loss(batch)
    np.random.seed(123)
    B, S, N, _, H, W = batch[0].shape
    flips = np.random.uniform(size=(B * S * N)) > 0.5
    assert flips.any() and not flips.all()
    assert np.array_equal(seen["flipped"].cpu().numpy().astype(bool), flips)
    # the composition of the reference's steps, each through its own function
    labels = ts.get_depth_labels(batch[0], batch[1], batch[2])
    norm = ts.normalize_images(batch[0])
    fH, fW = H // ts.downsample, W // ts.downsample
    np.random.seed(123)
    imgs2, labels2, flips2 = ts.augment_images(norm, labels.view(B * S * N, fH, fW, -1), 'train')
    assert np.array_equal(flips2, flips)
    assert torch.equal(seen["images"], imgs2)
    assert seen["oracle"] is not None and torch.equal(seen["oracle"], labels2.permute(0, 3, 1, 2))
    assert seen["oracle"].shape == (B * S * N, ts.depth_channels, fH, fW)
    assert "flipped" in batch[1] and not bool(batch[1]["flipped"].any())            # the batch's own dict was not touched
    # each branch matters
    losses = {}
    for aug, oracle in ((True, True), (False, True), (True, False)):
        ts.augment, ts.pass_depth_labels = aug, oracle
        np.random.seed(123)
        with torch.no_grad():
            losses[(aug, oracle)] = float(ts.forward_loss(batch)[0])
    assert abs(losses[(True, True)] - float(loss)) <= 1e-4 * abs(float(loss))
    assert losses[(False, True)] != losses[(True, True)] and losses[(True, False)] != losses[(True, True)]
