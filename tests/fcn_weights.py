"""Loads rcvpose_amd/synthetic_weights.py by file path (so that tests/golden/make_golden.py, which keeps the
repo root off sys.path while the reference is imported, can use the same generator)."""
import importlib.util
import os

_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rcvpose_amd", "synthetic_weights.py")
_spec = importlib.util.spec_from_file_location("_rcv_synthetic_weights", _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)

make_tensor = _mod.make_tensor
make_state_dict_numpy = _mod.make_state_dict_numpy
make_state_dict = _mod.make_state_dict
make_input = _mod.make_input
