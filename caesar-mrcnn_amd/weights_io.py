"""Weight files: tensors keyed "<layer>/<weight>" in the Keras layouts (Conv2D HWIO, Dense [in,out],
Conv2DTranspose (2,2,out,in), BatchNorm gamma/beta/moving_mean/moving_variance).

``.npz`` is the native container.  ``.h5`` is the Keras-HDF5 layout the reference reads and writes
(mrcnn/model.py:2197-2239, 2461-2462): see hdf5_min.py for the dependency-free subset reader/writer
(h5py is not available on the target image).
"""
import os

import numpy as np


def save(path, tensors, layout=None):
    d = os.path.dirname(os.path.abspath(path))
    if d and not os.path.exists(d):
        os.makedirs(d)
    if path.endswith(".h5"):
        from . import hdf5_min
        hdf5_min.save_keras_weights(path, tensors, layout)
    else:
        with open(path if path.endswith(".npz") else path + ".npz", "wb") as f:
            np.savez(f, **{k.replace("/", "|"): np.asarray(v) for k, v in tensors.items()})


def load(path):
    if path.endswith(".h5"):
        from . import hdf5_min
        return hdf5_min.load_keras_weights(path)
    with np.load(path, allow_pickle=False) as z:
        return {k.replace("|", "/"): z[k] for k in z.files}
