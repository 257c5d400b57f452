#!/opt/conda/bin/python3.9
"""Writes tests/golden/sr_data_*.hdf5 with the real h5py (3.3.0, under /opt/conda in this image; the reference pins
3.6.0 -- same default file format) using exactly the calls of the reference's writer
(superresolution_scripts/augmentation_utils.py:117-136): create_dataset(data=<list of [h,w,1] float32 arrays>),
create_dataset("angles"/"shifts"), attrs filename / mode (str) and angle_max / shift_max (numbers).
Run: /opt/conda/bin/python3.9 tests/golden/make_hdf5_golden.py   (not needed at test time: the files are committed)."""
import os
import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.RandomState(7)


def write(name, mode, n, h, w):
    class_masks = [rng.rand(h, w, 1).astype(np.float32) * (8.0 if mode == "argmax" else 1.0) for _ in range(n)]
    max_masks = [rng.rand(h, w, 1).astype(np.float32) for _ in range(n)]
    angles = rng.uniform(-0.15, 0.15, n).astype(np.float32)
    shifts = rng.uniform(-80, 80, (n, 2)).astype(np.float32)
    angles[0] = 0
    shifts[0] = 0
    f = h5py.File(os.path.join(here, name), "w")
    f.create_dataset("class_masks", data=class_masks)
    if mode == "slice_max":
        f.create_dataset("max_masks", data=max_masks)
    f.create_dataset("angles", data=angles)
    f.create_dataset("shifts", data=shifts)
    f.attrs["filename"] = "2007_000033"
    f.attrs["mode"] = mode
    f.attrs["angle_max"] = 0.15
    f.attrs["shift_max"] = 80
    f.close()
    np.savez(os.path.join(here, name.replace(".hdf5", "_expected.npz")), class_masks=np.stack(class_masks),
             max_masks=np.stack(max_masks) if mode == "slice_max" else np.zeros(0, np.float32), angles=angles, shifts=shifts)


def write_keras_like(name):
    """The layout of keras Model.save_weights(.h5) (keras/saving/hdf5_format.py save_weights_to_hdf5_group): root attrs
    layer_names / backend / keras_version, one group per layer with attr weight_names and datasets <layer>/<var>:0."""
    layers = {"entry_flow_conv1_1": {"kernel:0": (3, 3, 3, 4)},
              "entry_flow_conv1_1_BN": {"gamma:0": (4,), "beta:0": (4,), "moving_mean:0": (4,), "moving_variance:0": (4,)},
              "entry_flow_block1_separable_conv1_depthwise": {"depthwise_kernel:0": (3, 3, 4, 1)},
              "entry_flow_block1_separable_conv1_pointwise": {"kernel:0": (1, 1, 4, 8)},
              "logits_semantic": {"kernel:0": (1, 1, 8, 3), "bias:0": (3,)},
              "activation_without_weights": {}}
    for i in range(12):                                   # more than one symbol-table node in the root group
        layers[f"middle_flow_unit_{i + 1}_separable_conv1_pointwise"] = {"kernel:0": (1, 1, 2, 2)}
    f = h5py.File(os.path.join(here, name), "w")
    f.attrs["layer_names"] = [n.encode("utf8") for n in layers]
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["keras_version"] = "2.7.0".encode("utf8")
    expected = {}
    for lname, ws in layers.items():
        g = f.create_group(lname)
        g.attrs["weight_names"] = [f"{lname}/{w}".encode("utf8") for w in ws]
        for w, shape in ws.items():
            val = rng.standard_normal(shape).astype(np.float32)
            g.create_dataset(f"{lname}/{w}", data=val)
            expected[f"{lname}/{w.split(':')[0]}"] = val
    f.close()
    np.savez(os.path.join(here, name.replace(".h5", "_expected.npz")), **expected)


def write_unsupported(name):
    f = h5py.File(os.path.join(here, name), "w")
    f.create_dataset("chunked", data=np.arange(64, dtype=np.float32).reshape(8, 8), chunks=(4, 4), compression="gzip")
    f.close()


write_keras_like("keras_like_weights.h5")
write_unsupported("chunked_gzip.hdf5")
write("sr_data_argmax.hdf5", "argmax", 5, 6, 4)
write("sr_data_slice_max.hdf5", "slice_max", 4, 3, 5)
print("written")
