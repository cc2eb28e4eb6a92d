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


def __getattr__(name):
    # lazy: importing the package must not need torch.cuda / the HIP library
    if name == 'OpticalFlow':
        from .of import OpticalFlow
        return OpticalFlow
    raise AttributeError(name)
