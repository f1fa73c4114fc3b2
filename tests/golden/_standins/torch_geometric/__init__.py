"""Stand-in for the absent `torch_geometric` package (see ../README.md)."""
