"""Per-epoch seeded shuffling of the training ids -- defines the seed order the sampler sees
(reference: fast_trainer/shufflers.py:6-45: seed = 2147483647 + epoch, randperm on the ids' device,
per-rank contiguous slice)."""
import torch


class Shuffler:
    DEFAULT_INITIAL_SEED = 2147483647

    def __init__(self, idx: torch.Tensor, initial_seed: int = DEFAULT_INITIAL_SEED):
        assert idx.dim() == 1
        self.initial_idx = idx
        self.initial_seed = initial_seed
        self.generator = torch.Generator(device='cpu')
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def get_idx(self):
        self.generator.manual_seed(self.initial_seed + self.epoch)
        # the permutation is always drawn by the CPU generator (what the reference does for its
        # host-resident ids), so a given (seed, epoch) orders the ids identically wherever they live
        perm = torch.randperm(self.initial_idx.numel(), generator=self.generator)
        return self.initial_idx[perm.to(self.initial_idx.device)]


class DistributedShuffler(Shuffler):
    def __init__(self, idx, world_size, initial_seed=Shuffler.DEFAULT_INITIAL_SEED):
        super().__init__(idx, initial_seed)
        self.world_size = world_size

    def get_idx(self, rank):
        shuffled = super().get_idx()
        n = shuffled.numel()
        return shuffled[(n * rank) // self.world_size:(n * (rank + 1)) // self.world_size]


class FederatedDistributedShuffler(Shuffler):
    """Each rank shuffles the training ids of its own partition (shufflers.py:92-101)."""
