"""One-time conversion of a MATLAB v7.3 (HDF5) cell array of kernels (the reference's kernels/Levin09.mat) to <name>.npz with keys k0, k1, ...,
for interpreters without h5py (diffpir_amd/main_ddpir.py::load_reference_kernels reads it).  Run under a Python that HAS h5py, e.g.
    /opt/conda/bin/python3.9 tools/convert_mat_v73.py /path/to/kernels/Levin09.mat
h5py sees MATLAB's column-major arrays transposed; hdf5storage.loadmat (what the reference calls) un-transposes, and so does this."""
import os
import sys

import h5py
import numpy as np

path = sys.argv[1]
with h5py.File(path, "r") as f:
    refs = f["kernels"]
    ks = {f"k{i}": np.array(f[refs[i, 0]]).T for i in range(refs.shape[0])}
out = os.path.splitext(path)[0] + ".npz"
np.savez(out, **ks)
print("wrote", out, {k: v.shape for k, v in ks.items()})
