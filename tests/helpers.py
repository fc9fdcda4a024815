"""Shared builders for the GPU tests: native models loaded with the synthetic weight recipe through the reference's
state-dict keys (dsml_thesis_amd/synth.py; oracle/weights.py holds the oracle's own copy of the recipe)."""
from dsml_thesis_amd.synth import (fr_config, load_recipe, make_fr_model, make_tf_model, make_uncond_model, tf_config, uncond_config)  # noqa: F401
