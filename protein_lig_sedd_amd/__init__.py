"""protein_lig_sedd_amd — MI355X-native SEDD hot path (HIP kernels behind a C ABI, drop-in host modules).

Importing the package does not touch the GPU; the first op loads libsedd_hip.so and raises if it is missing."""
__version__ = "0.1.0"
