from .general import (load_dataset, load_file, pad_nested_sequences, pad_sequence, prepare_input_from_neighbors, prepare_input_pmt,
                      process_xyz_pmt, split_data)
from .datagenerator import DataIterator
from .packed_dataset import PackedDataset
from .voronoi_neighbor import compute_voronoi_neighbor

__all__ = ["DataIterator", "PackedDataset", "compute_voronoi_neighbor", "load_dataset", "load_file", "pad_nested_sequences", "pad_sequence",
           "prepare_input_from_neighbors", "prepare_input_pmt", "process_xyz_pmt", "split_data"]
