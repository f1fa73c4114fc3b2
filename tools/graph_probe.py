#!/usr/bin/env python3
"""Which part of the step survives hipGraph replay?  python tools/graph_probe.py <stage> [replays]
stages: build (relation build only) | geom (+ edge geometry + embedding) | fwd (no-grad forward) | step (energy + forces).
Every stage: 3 eager warm-ups on a side stream, capture, then replays interleaved with eager runs of the same stage;
outputs must stay bit-identical to the first eager result."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402
from hermnet_amd.ops import EdgeGeometry  # noqa: E402
from hermnet_amd.relations import RelationalGraph  # noqa: E402


def main():
    stage = sys.argv[1]
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    reps = (4, 4, 8) if os.environ.get("PROBE_SMALL", "0") != "0" else (10, 10, 25)
    d = synth.fcc_alloy(reps=reps, device=dev)
    model = hn.HVNet(["Al", "Ni", "Cu"], rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
    model = model.to(dev)
    for p in model.parameters():
        p.requires_grad_(False)
    zl = [13, 28, 29]

    def run():
        if stage == "build":
            g = RelationalGraph.build(d.atomic_number, d.edge_index, zl, edge_shift=d.edge_shift, batch=d.batch)
            return [g.csr_src.clone(), g.csc_pos.clone(), g.csc_rowptr.clone(), g.row_active.clone()]   # (no reductions)
        if stage == "geom":
            g = RelationalGraph.build(d.atomic_number, d.edge_index, zl, edge_shift=d.edge_shift, batch=d.batch)
            e = EdgeGeometry.apply(d.pos.detach(), d.cell, g)
            x = torch.nn.functional.embedding(g.z_rows, model.embed.weight)
            return [e.clone(), x.clone()]
        if stage == "fwd":
            with torch.no_grad():
                return [model(d)]
        d.pos.requires_grad_(True)
        en = model(d)
        f = -torch.autograd.grad(en.sum(), d.pos)[0]
        return [en, f]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ref = [t.detach().clone() for t in run()]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(stage, "eager ok", [float(t.double().sum()) for t in ref], flush=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = run()
    torch.cuda.synchronize()
    print(stage, "captured", flush=True)
    for i in range(replays):
        graph.replay()
        torch.cuda.synchronize()
        ok = all(torch.equal(a.detach(), b) for a, b in zip(out, ref))
        print(stage, "replay", i, "bit-identical" if ok else "DIFFERENT", flush=True)
        eag = [t.detach().clone() for t in run()]          # an eager run in between
        torch.cuda.synchronize()
        if os.environ.get("PROBE_NOCHECK", "0") != "0":
            continue                                        # (partial builds leave outputs uninitialised)
        assert all(torch.equal(a, b) for a, b in zip(eag, ref)), "eager result changed"
        if not ok:
            raise SystemExit(1)
    print(stage, "PASS", flush=True)


if __name__ == "__main__":
    main()
