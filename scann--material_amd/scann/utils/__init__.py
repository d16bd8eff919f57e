from .general import load_dataset, pad_nested_sequences, pad_sequence, split_data
from .datagenerator import DataIterator
from .packed_dataset import PackedDataset

__all__ = ["DataIterator", "PackedDataset", "load_dataset", "pad_nested_sequences", "pad_sequence", "split_data"]
