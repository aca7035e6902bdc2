"""Top-level alias of mm_training_amd.ops under the reference's package name, so the reference's own import lines
(`from ops.voxel_pooling import voxel_pooling`, layers/backbones/lss_fpn.py:11 and test/test_ops/test_voxel_pooling.py:5)
resolve to the HIP-backed op without editing them.  No code lives here."""
