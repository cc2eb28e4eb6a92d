"""Plugin loader with the reference's behaviour (utils/model.py:10-47):
``<flownet_path.name>.net`` is imported BY NAME from sys.path, the kwargs of
``options2model_kwargs`` are filtered by ``Model.__init__``'s signature, an
optional ``--sp`` state dict is loaded, the model is moved to the device.
``--flownet_path dvs_of_training_framework_amd`` selects the HIP build;
any other package that honours the contract (e.g. DummyNet) still loads."""
import importlib
import importlib.util
import inspect
import logging
from pathlib import Path

import torch

from .options import options2model_kwargs


def filter_kwargs(func, kwargs):
    signature = inspect.signature(func)
    keys2use = []
    for key in signature.parameters:
        # a **kwargs parameter accepts everything
        if signature.parameters[key].kind == inspect.Parameter.VAR_KEYWORD:
            return kwargs
        if key in kwargs:
            keys2use.append(key)
    keys_not2use = [k for k in kwargs if k not in signature.parameters]
    if len(keys_not2use):
        logging.warning(f'{keys_not2use} are filtered out from '
                        'OpticalFlow parameters!')
    return {key: kwargs[key] for key in keys2use}


def import_module(module_name, module_path):
    module_spec = importlib.util.find_spec(module_name)
    assert module_spec is not None, f'Module: {module_name} at ' \
        f'{Path(module_path).resolve()} not found'
    return importlib.import_module(module_name)


def init_model(args, device):
    module = import_module(f'{Path(args.flownet_path).name}.net',
                           Path(args.flownet_path) / 'net.py')
    model_kwargs = options2model_kwargs(args)
    model_kwargs = filter_kwargs(module.Model, model_kwargs)
    model = module.Model(device, **model_kwargs)
    if getattr(args, 'sp', None) is not None:
        state_dict = torch.load(args.sp, map_location=device,
                                weights_only=True)
        if 'model' in state_dict:
            state_dict = state_dict['model']
        model.load_state_dict(state_dict)
    model.to(device)
    return model
