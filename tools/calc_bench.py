#!/usr/bin/env python3
"""ms per `NNCalculator.calculate` call (what an ASE MD loop pays per step: upload, neighbour search, step, results to the
host), eager against `graph_replay=True`, on fcc alloy cells of growing size:   python tools/calc_bench.py [calls]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402
from hermnet_amd.elements import chemical_symbols  # noqa: E402
from hermnet_amd.plugin import NNCalculator  # noqa: E402


class Atoms(object):            # duck-typed stand-in for ase.Atoms (ASE is not installed on this image)
    def __init__(self, pos, z, cell, symbols=None):
        self.positions, self.cell, self.pbc = pos, cell, [True] * 3
        self.numbers = z        # (ase.Atoms keeps the atomic numbers as an array)
        self._sym = symbols if symbols is not None else [chemical_symbols[int(v)] for v in z]

    def get_chemical_symbols(self):
        return self._sym


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
    for reps in [(2, 2, 4), (3, 3, 6), (5, 5, 10), (10, 10, 25)]:
        pos, cell, z = synth.fcc_alloy_atoms(reps=reps, seed=0)
        rs = np.random.RandomState(0)
        out = []
        for replay in (False, True):
            model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval()
            model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
            calc = NNCalculator(model, None, trn_mean=0.0, device_="cuda:0", graph_replay=replay)
            p = pos.copy()
            sym = [chemical_symbols[int(v)] for v in z]
            for _ in range(5):
                calc.calculate(Atoms(p, z, cell, sym), ["energy", "forces"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(calls):
                p = p + rs.normal(scale=0.002, size=p.shape)
                calc.calculate(Atoms(p, z, cell, sym), ["energy", "forces"])
            out.append((time.perf_counter() - t0) / calls * 1e3)
            e = calc.results["energy"]
        print("atoms %6d   eager %7.3f ms/call   replayed %7.3f ms/call   x%.2f   (E %.4f)" % (
            len(z), out[0], out[1], out[0] / out[1], e), flush=True)


if __name__ == "__main__":
    main()
