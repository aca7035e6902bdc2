"""Top-level alias of mm_training_amd.models (the reference keeps models/ as a plain directory with bev_depth.py)."""
