"""Deterministic latent datasets -- the seed contract of utils/zdataset.py: z number i of
``standard_z_sample(n, depth, seed)`` is the same for every n, drawn from
``numpy.random.RandomState(seed).standard_normal`` (:37-51)."""
import numpy
import torch
from torch.utils.data import TensorDataset


def standard_z_sample(size, depth, seed=1, device=None):
    rng = numpy.random.RandomState(seed)
    z = torch.from_numpy(rng.standard_normal(size * depth).reshape(size, depth)).float()
    return z if device is None else z.to(device)


def standard_y_sample(size, num_classes, seed=1, device=None):
    rng = numpy.random.RandomState(seed)
    y = torch.from_numpy(rng.randint(num_classes, size=size)).long()
    return y if device is None else y.to(device)


def z_sample_for_model(model, size=100, seed=1):
    """Shape follows ``model.input_shape`` if present, else the model's first
    conv/linear layer: (size, C, 1, 1) for convolutional inputs, (size, C) for linear."""
    if hasattr(model, 'input_shape'):
        return standard_z_sample(size, model.input_shape[1], seed=seed).view(
            (size,) + tuple(model.input_shape[1:]))
    conv_types = (torch.nn.Conv2d, torch.nn.ConvTranspose2d)
    for m in model.modules():
        if isinstance(m, conv_types):
            return standard_z_sample(size, m.in_channels, seed=seed)[:, :, None, None]
        if isinstance(m, torch.nn.Linear):
            return standard_z_sample(size, m.in_features, seed=seed)
    raise ValueError('model has no conv or linear layer to infer the latent size from')


def z_dataset_for_model(model, size=100, seed=1, indices=None):
    if indices is None:
        return TensorDataset(z_sample_for_model(model, size, seed))
    indices = torch.as_tensor(indices, dtype=torch.int64, device='cpu')
    return TensorDataset(z_sample_for_model(model, indices.max().item() + 1, seed)[indices])
