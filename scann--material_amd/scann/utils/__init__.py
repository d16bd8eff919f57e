from .general import load_dataset, pad_nested_sequences, pad_sequence, split_data
from .datagenerator import DataIterator

__all__ = ["DataIterator", "load_dataset", "pad_nested_sequences", "pad_sequence", "split_data"]
