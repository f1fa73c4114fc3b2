def primitive_neighbor_list(*a, **k):  # import-only placeholder (HermNet/data.py:11)
    raise RuntimeError("ase is not available; the generator builds neighbour lists itself")
