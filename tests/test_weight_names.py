"""Real-weights readiness without the file (reference model.py:9,129-145: the bonlime checkpoint is loaded BY NAME with
skip_mismatch, so a name this framework does not expect would be dropped silently, layer by layer).

tests/golden/keras_variables.json holds every Keras variable name and shape of the reference's graph, derived from the
naming rules of model.py:149-541 by tests/golden/make_keras_names.py (channel widths propagated through the graph, and
cross-checked there against the layer-name literals in the reference's text).  Here: the loader expects exactly that set.
"""
import json
import os

import numpy as np
import pytest

from asr_amd import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = [("xception", "xception", 21), ("mobilenet", "mobilenet", 21), ("xception_5_classes", "xception", 5)]


@pytest.fixture(scope="module")
def keras_variables():
    with open(os.path.join(GOLDEN, "keras_variables.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("key,backbone,classes", CASES)
def test_the_loader_expects_exactly_the_reference_graphs_variables(keras_variables, key, backbone, classes):
    expected = {k: tuple(v) for k, v in keras_variables[key].items()}
    template = W.make_synthetic_weights(7, classes, backbone)
    got = {k: tuple(v.shape) for k, v in template.items()}
    assert set(got) == set(expected), sorted(set(got) ^ set(expected))[:10]
    assert got == expected
    if key == "xception":      # Keras' summary of this graph: 41 258 213 parameters in all (with the BatchNorm statistics)
        assert sum(int(np.prod(s)) for s in expected.values()) == 41258213 == W.count_params()


def test_a_checkpoint_with_exactly_those_variables_is_taken_whole(keras_variables):
    """merge_by_name (Keras by_name + skip_mismatch, per layer): nothing skipped, every value from the file."""
    expected = keras_variables["mobilenet"]
    template = W.make_synthetic_weights(7, 21, "mobilenet")
    rng = np.random.default_rng(0)
    loaded = {k: rng.standard_normal(s).astype(np.float32) for k, s in expected.items()}
    merged, skipped = W.merge_by_name(template, loaded)
    assert skipped == [] and W.loaded_fraction(template, skipped) == 1.0
    assert all(np.array_equal(merged[k], loaded[k]) for k in expected)


def test_one_renamed_or_reshaped_variable_is_reported_not_swallowed(keras_variables):
    """The failure this guards against: a silent per-layer fall back to the initial values."""
    expected = keras_variables["xception"]
    template = {k: np.zeros(s, np.float32) for k, s in expected.items()}
    loaded = {k: np.ones(s, np.float32) for k, s in expected.items()}
    loaded["aspp1_depthwise/kernel"] = loaded.pop("aspp1_depthwise/depthwise_kernel")          # wrong variable name
    loaded["decoder_conv0_pointwise/kernel"] = np.ones((1, 1, 256, 256), np.float32)          # 304 input channels expected
    merged, skipped = W.merge_by_name(template, loaded)
    names = {n for n, _ in skipped}
    assert names == {"aspp1_depthwise/depthwise_kernel", "decoder_conv0_pointwise/kernel"}
    assert merged["aspp1_depthwise/depthwise_kernel"].sum() == 0 and merged["aspp1_pointwise/kernel"].min() == 1.0
    assert 0.98 < W.loaded_fraction(template, skipped) < 1.0


def test_keras_h5_dataset_paths_of_those_names_map_back(keras_variables):
    """A Keras weight file stores variable v of layer L as dataset  L/L/v:0  (model.save_weights; the bonlime file's layout;
    full-model files put the same tree under model_weights/).  load_weights maps every such path back to "L/v"; the HDF5
    reading itself is pinned by tests/golden/keras_like_weights.h5, written with the real h5py (tests/test_hdf5_lite.py)."""
    for key in ("xception", "mobilenet", "xception_5_classes"):
        for name in keras_variables[key]:
            layer, var = name.split("/")
            assert W.keras_variable_name(f"{layer}/{layer}/{var}:0") == name
            assert W.keras_variable_name(f"model_weights/{layer}/{layer}/{var}:0") == name
    assert W.keras_variable_name("optimizer_weights/Adam/iterations:0") is None
    assert W.keras_variable_name("top_level_dataset") is None
