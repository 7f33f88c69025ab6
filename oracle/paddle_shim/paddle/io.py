"""ORACLE (test infrastructure) -- import-only stand-ins for ``paddle.io`` (the reference's normalizer / reader modules
import these at load time; the encoder path never uses them)."""


class Dataset:
    pass


class IterableDataset:
    pass


class Sampler:
    def __init__(self, data_source=None):
        self.data_source = data_source


class BatchSampler(Sampler):
    def __init__(self, *a, **k):
        raise NotImplementedError("paddle shim: data loading is out of scope")


class DistributedBatchSampler(BatchSampler):
    pass


class DataLoader:
    def __init__(self, *a, **k):
        raise NotImplementedError("paddle shim: data loading is out of scope")
