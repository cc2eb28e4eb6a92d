"""MI355X-native optical-flow training hot path of
e-sha/dvs_of_training_framework: a drop-in ``--flownet_path`` package
(``net.Model``, ``OpticalFlow``) plus HIP-backed counterparts of
``utils.loss`` / ``utils.training`` / ``utils.model`` / ``utils.options``."""


def __getattr__(name):
    # lazy: importing the package must not need torch.cuda / the HIP library
    if name == 'OpticalFlow':
        from .of import OpticalFlow
        return OpticalFlow
    raise AttributeError(name)
