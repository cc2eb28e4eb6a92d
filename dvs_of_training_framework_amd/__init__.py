"""MI355X-native optical-flow training hot path of
e-sha/dvs_of_training_framework: a drop-in ``--flownet_path`` package
(``net.Model``, ``OpticalFlow``) plus HIP-backed counterparts of
``utils.loss`` / ``utils.training`` / ``utils.model`` / ``utils.options``."""
import os

# The backward runs on two HIP streams (predictor.py).  ROCclr multiplexes
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); once RCCL has
# made its own streams the second backward stream can land on the SAME hardware
# queue as the main stream, and the two then alternate instead of overlapping
# (measured: 4.78 vs 4.36 ms/step with a process group).  Read when the HIP
# runtime initialises, i.e. at the first torch.cuda use -- after this import.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
# A captured training step (capture.py) replays its two backward branches on
# the graph executor's own queues.  Measured on ROCm 7.2 (bench.py --graph,
# batch 8): default queue count 4.17 ms/step (the branches spread over four
# queues and the MFMA kernels slow each other down: 5.1 ms of kernel time per
# step), 2 queues 3.52 ms, 1 queue 3.57 ms -- the eager two-stream schedule is
# 3.25 ms.  Read by the HIP runtime at initialisation.
os.environ.setdefault('DEBUG_HIP_FORCE_GRAPH_QUEUES', '2')


def __getattr__(name):
    # lazy: importing the package must not need torch.cuda / the HIP library
    if name == 'OpticalFlow':
        from .of import OpticalFlow
        return OpticalFlow
    raise AttributeError(name)
