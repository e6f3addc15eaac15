import torch


def load(file, model, optimizer=None, map_location='cpu'):
    state = torch.load(file, map_location=map_location, weights_only=False)
    model.load_state_dict(state['model'])
    rest = {k: v for k, v in state.items() if k not in ('model', 'optimizer')}
    return model, optimizer, rest


def save(*a, **k):
    raise NotImplementedError


def latest_path(*a, **k):
    return None


def best_path(*a, **k):
    return None
