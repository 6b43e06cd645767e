"""Fixed index samplers used by tally.make_loader (reference: utils/sampler.py:20-48)."""
from torch.utils.data.sampler import Sampler


class FixedSubsetSampler(Sampler):
    def __init__(self, samples):
        self.samples = list(samples)

    def __iter__(self):
        return iter(self.samples)

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, key):
        return self.samples[key]

    def dereference(self, indices):
        return [self.samples[i] for i in indices]

    def subset(self, new_subset):
        return FixedSubsetSampler(self.dereference(new_subset))
