"""Stand-in for the absent `ase` package: element table and import-only placeholders."""
