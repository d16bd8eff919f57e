from .general import (load_dataset, pad_nested_sequences, pad_sequence, prepare_input_from_neighbors, prepare_input_pmt,
                      split_data)
from .datagenerator import DataIterator
from .packed_dataset import PackedDataset

__all__ = ["DataIterator", "PackedDataset", "load_dataset", "pad_nested_sequences", "pad_sequence", "prepare_input_from_neighbors",
           "prepare_input_pmt", "split_data"]
