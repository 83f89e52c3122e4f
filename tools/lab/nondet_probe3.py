import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tests.test_parity_gates_gpu as T
for cfg in sys.argv[1:] or ["2d", "3d"]:
    try:
        T.test_full_size_graph_path_properties(cfg)
        print(cfg, "passed", flush=True)
    except AssertionError as e:
        import traceback; traceback.print_exc(limit=2)
        print(cfg, "FAILED", flush=True)
