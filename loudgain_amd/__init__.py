"""loudgain_amd -- MI355X-native EBU R128 loudness scanner behind loudgain's scan.c API.

Layout: csrc/ holds the HIP kernels and the C-ABI shared library
(libloudscan_hip.so); `device` is the ctypes binding of the device-level ABI
(include/loudscan_device.h); `scan` mirrors /root/reference/src/scan.h;
`album` shards an album across GPUs (one process per GPU, RCCL via
torch.distributed); `synth` generates the benchmark program material.
There is no CPU fallback: every scan entry point raises if the HIP library or a
GPU is missing.
"""
__version__ = "0.1.0"
