"""Run by tests/test_gpu_parity.py::test_alternative_kernel_variants in a child process: the kernel variant /
split mode is read from the environment once per process (csrc/message_kernels.hip), so every non-default
variant needs a process of its own.  Checks energies + forces of four golden cases (H = 128 and 256) at 1e-5."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402

from helpers import Golden, rel_err  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for name in ["c1_si64", "alloy108", "mol16", "alloy32_h256"]:
        g = Golden(name)
        model = g.model().to(dev)
        d = g.data().to(dev)
        d.pos.requires_grad_(True)
        e = model(d)
        f = -torch.autograd.grad(e.sum(), d.pos)[0]
        ee, fe = rel_err(e.detach().cpu(), g.energy), rel_err(f.cpu(), g.forces)
        print("%s: rel err E %.2e F %.2e" % (name, ee, fe))
        assert ee < 1e-5 and fe < 1e-5, name
    print("VARIANT_OK")


if __name__ == "__main__":
    main()
