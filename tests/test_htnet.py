"""HTNet (BASELINE configs[2]).  The reference's class is a stub (`HermNet/hermnet.py:155-157` raises
NotImplementedError), so there is nothing of the reference to pin against: PARITY UNPINNED.  The model is
build-defined (DESIGN.md "HTNet"; figs/subgraph.svg (c), README.md:27); these tests check
  * the product path against the CPU oracle of that specification (`oracle.htnet_energy`),
  * what any such model must satisfy whatever its weights: rotation / translation / permutation invariance of the
    energy, equivariance of the forces, extensivity, Newton's third law,
  * the degenerate case T = 1, where HTNet must BE the (reference-pinned) HVNet.
"""
import math

import numpy as np
import pytest
import torch

import hermnet_amd as hn
import ref_ops
from helpers import Golden, rel_err
from hermnet_amd import synth
from oracle import hermnet_oracle as orc

KW = dict(rc=5.0, num_layers=2, hidden_channels=64, num_rbf=32)


def _model(elems, kw=KW, seed=7):
    m = hn.HTNet(elems, **kw).eval()
    sd = synth.synth_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    for p in m.parameters():
        p.requires_grad_(False)
    return m, sd


def _cpu_ops(monkeypatch):
    import hermnet_amd.hermnet as hmod
    import hermnet_amd.layer as lmod
    monkeypatch.setattr(hmod.HVNet, "_require_device", staticmethod(lambda pos: None))
    monkeypatch.setattr(hmod, "EdgeGeometry", ref_ops.RefEdgeGeometry)
    for fn in ["energy_head_fwd", "energy_head_bwd", "layernorm_fwd", "layernorm_bwd", "ssilu_fwd", "ssilu_bwd",
               "update_mid", "update_out", "update_out_bwd", "update_mid_bwd", "node_pre_fwd", "node_pre_bwd",
               "node_update_fwd", "node_update_bwd", "node_update_pre_fwd", "node_pre_fwd16", "node_pre_bwd16"]:
        monkeypatch.setattr(lmod.nodeops, fn, getattr(ref_ops, fn))
    monkeypatch.setattr(lmod, "_msg_fwd", ref_ops.msg_fwd)
    monkeypatch.setattr(lmod, "_msg_bwd", ref_ops.msg_bwd)


def _ef(model, d):
    d.pos.requires_grad_(True)
    e = model(d)
    f = -torch.autograd.grad(e.sum(), d.pos)[0]
    return e.detach(), f


def test_relation_keys_and_state_dict_layout():
    keys = [k for k, _, _ in hn.hermnet.triadic_relations(["Al", "Ni", "Cu"])]
    assert len(keys) == 18 and keys[0] == "Al_Al-Al" and keys[5] == "Al_Cu-Cu" and keys[-1] == "Cu_Cu-Cu"
    assert keys == [k for k, _, _ in orc.triadic_relations(["Al", "Ni", "Cu"])]
    m = hn.HTNet(["Al", "Ni", "Cu"], num_layers=2, hidden_channels=64, num_rbf=16)
    sd = m.state_dict()
    assert "hermconvs.1.mods.Ni_Al-Cu.message_layer.rbf_proj.weight" in sd and "embed.weight" in sd
    assert sum(k.startswith("hermconvs.0.mods.") for k in sd) == 18 * 13     # 13 tensors per PaiNNModule, as in HVNet


@pytest.mark.parametrize("name,elems", [("alloy108", ["Al", "Ni", "Cu"]), ("alloy108_unknown_type", ["Al", "Ni"]),
                                        ("mol16", ["H", "C", "O"]), ("c1_si64", ["Si"])])
def test_triadic_graph_structure(name, elems):
    """Every expanded edge sits in a relation whose pair contains its source's element and whose centre is its
    target's element; each original edge between known elements appears exactly T times; CSR / CSC are consistent."""
    from hermnet_amd.relations import RelationalGraph
    from hermnet_amd.elements import atomic_numbers
    g = Golden(name)
    d = g.data()
    zl = [atomic_numbers[e] for e in elems]
    gr = RelationalGraph.build_triadic(d.atomic_number, d.edge_index, zl, d.get("edge_shift"), d.batch)
    T = len(zl)
    pairs = [(p, q) for p in range(T) for q in range(p, T)]
    P, B = len(pairs), gr.block
    assert gr.T == T * P and gr.N == gr.T * B and gr.triadic_pairs == P
    z = d.atomic_number
    known = torch.tensor([int(v) in zl for v in z])
    n_known_edges = int((known[d.edge_index[0]] & known[d.edge_index[1]]).sum())
    assert gr.E == T * n_known_edges
    rowptr = gr.csr_rowptr.long()
    tgt_row = torch.repeat_interleave(torch.arange(gr.N), rowptr[1:] - rowptr[:-1])
    rel = tgt_row // B
    zi, zj = z[gr.tgt_id.long()], z[gr.src_id.long()]
    for e in range(0, gr.E, max(1, gr.E // 500)):
        c, (p, q) = int(rel[e]) // P, pairs[int(rel[e]) % P]
        assert int(zi[e]) == zl[c] and int(zj[e]) in (zl[p], zl[q])
        assert int(gr.res_row[tgt_row[e]]) == int(gr.row_of_node[gr.tgt_id[e].long()])
        assert int(gr.csr_src[e]) == int(gr.row_of_node[gr.src_id[e].long()])
    # CSC: sorted by (relation, source row), consistent with CSR
    pos = gr.csc_pos.long()
    assert torch.equal(gr.csc_tgt.long(), tgt_row[pos])
    key = rel[pos] * gr.num_src + gr.csr_src.long()[pos]
    assert bool((key[1:] >= key[:-1]).all())
    assert int(gr.csc_rowptr[-1]) == gr.E and gr.csc_rowptr.numel() == gr.T * gr.num_src + 1


@pytest.mark.parametrize("name,elems", [("alloy108", ["Al", "Ni", "Cu"]), ("alloy108_unknown_type", ["Al", "Ni"]),
                                        ("mol16", ["H", "C", "O"])])
def test_host_pipeline_matches_oracle_cpu(name, elems, monkeypatch):
    """Row bookkeeping of the two row spaces, virtual-target residuals, the 1/P combine and the hand-written
    backward, with the kernels replaced by their PyTorch restatements."""
    _cpu_ops(monkeypatch)
    g = Golden(name)
    model, sd = _model(elems)
    e_ref, f_ref = orc.htnet_energy_and_forces(sd, elems, g.data(), **KW)
    e, f = _ef(model, g.data())
    assert rel_err(e, e_ref) < 5e-6 and rel_err(f, f_ref) < 1e-5


def test_single_element_htnet_is_hvnet_cpu(monkeypatch):
    """T = 1: one relation (Si; Si-Si) = all edges, P = 1 -> identical to HVNet, whose numerics ARE pinned against
    the reference (tests/golden/c1_si64.npz)."""
    _cpu_ops(monkeypatch)
    g = Golden("c1_si64")
    hv = g.model()
    ht = hn.HTNet(["Si"], **g.model_kw).eval()
    ht.load_state_dict({k.replace("mods.Si.", "mods.Si_Si-Si."): v for k, v in hv.state_dict().items()})
    for p in ht.parameters():
        p.requires_grad_(False)
    e, f = _ef(ht, g.data())
    assert rel_err(e, g.energy) < 2e-6 and rel_err(f, g.forces) < 1e-5


def test_training_mode_is_refused():
    m = hn.HTNet(["Si"], num_layers=1, hidden_channels=64, num_rbf=16).train()
    with pytest.raises((NotImplementedError, RuntimeError)):
        m(Golden("c1_si64").data())


# ------------------------------------------------------------------------------------------- GPU
def _dev():
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("name,elems,kw", [
    ("alloy108", ["Al", "Ni", "Cu"], KW),
    ("alloy108", ["Al", "Ni", "Cu"], dict(rc=5.0, num_layers=3, hidden_channels=128, num_rbf=128)),
    # a basis wider than one LDS tile: virtual target rows through the two tap-row windows of both message kernels
    ("alloy108", ["Al", "Ni", "Cu"], dict(rc=5.0, num_layers=2, hidden_channels=128, num_rbf=200)),
    ("alloy108_unknown_type", ["Al", "Ni"], KW),
    ("mol16", ["H", "C", "O"], KW),
    ("mol16_intensive", ["H", "C", "O"], dict(KW, intensive=True)),
    ("c1_si64", ["Si"], KW)])
def test_hip_path_matches_oracle(name, elems, kw):
    """Tolerance 1e-5 relative (forces relative to max |F|), as for HVNet."""
    g = Golden(name)
    model, sd = _model(elems, kw)
    okw = {k: v for k, v in kw.items()}
    e_ref, f_ref = orc.htnet_energy_and_forces(sd, elems, g.data(), **okw)
    e, f = _ef(model.to(_dev()), g.data().to(_dev()))
    assert rel_err(e.cpu(), e_ref) < 1e-5 and rel_err(f.cpu(), f_ref) < 1e-5


def _htnet_training_check(name, elems, kw, device, monkeypatch=None):
    """One training step of `example/dist_train.py:86-99` (energy + force loss, create_graph=True, backward to every
    parameter) on HTNet in train() mode vs autograd through the oracle."""
    from test_training import training_step, assert_grads_close
    g = Golden(name)
    model, sd = _model(elems, kw)
    for p in model.parameters():
        p.requires_grad_(True)
    model = model.to(device).train()
    d = g.data()
    gen = torch.Generator().manual_seed(11)
    e0, _ = orc.htnet_energy_and_forces(sd, elems, d, **kw)
    y = e0 + 0.5 * torch.randn(e0.numel(), generator=gen)
    ftgt = 0.5 * torch.randn(d.pos.shape, generator=gen)
    lo, leo, lfo, og = orc.htnet_training_loss_and_grads(sd, elems, d, y, ftgt, 0.8, **kw)
    l, le, lf = training_step(model, g.data().to(device), y.to(device), ftgt.to(device), 0.8)
    scale = max(1.0, float(e0.abs().max())) * max(1.0, float(lo))
    assert abs(float(l) - float(lo)) < 2e-5 * scale and abs(float(lf) - float(lfo)) < 2e-5 * max(1.0, float(lfo))
    # relations without an edge are skipped (hermnet.py:56-57): their modules get no gradient on either side
    want = {k: v for k, v in og.items() if v is not None}
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None and k in want}
    extra = [k for k, p in model.named_parameters() if p.grad is not None and k not in want and float(p.grad.abs().max()) > 0]
    assert not extra, extra
    assert_grads_close(got, want)


@pytest.mark.parametrize("name,elems", [("alloy108", ["Al", "Ni", "Cu"]), ("mol16", ["H", "C", "O"])])
def test_train_mode_parameter_gradients_match_oracle_cpu(name, elems, monkeypatch):
    """HTNet.train(): the differentiable path on the triadic graph (two row spaces, residual through `res_row`, mean over
    a centre's pair relations), host tensors; every parameter gradient against the oracle's autograd."""
    import hermnet_amd.hermnet as hmod
    monkeypatch.setattr(hmod.HVNet, "_require_device", staticmethod(lambda pos: None))
    _htnet_training_check(name, elems, KW, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,elems,kw", [("alloy108", ["Al", "Ni", "Cu"], KW),
                                           ("mol16", ["H", "C", "O"], dict(rc=5.0, num_layers=3, hidden_channels=128, num_rbf=128))])
def test_train_mode_parameter_gradients_match_oracle_gpu(name, elems, kw):
    """The same step on the GPU: training kernels (csrc/train_kernels.hip), bucketed basis, native triadic build."""
    _htnet_training_check(name, elems, kw, _dev())


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["alloy108", "mol16", "c1_si64", "alloy10k", "molecules64"])
def test_native_triadic_build_is_the_torch_build(case, monkeypatch):
    """`hermnet_build_triadic` (counting sort over the expanded edge list) vs the torch-op build that defines the
    result: every array of the graph bit for bit, twice (the second call reuses the cached row layout)."""
    from hermnet_amd.relations import RelationalGraph
    from hermnet_amd.elements import atomic_numbers
    dev = _dev()
    if case == "alloy10k":
        d, elems = synth.fcc_alloy(), ["Al", "Ni", "Cu"]
    elif case == "molecules64":
        d, elems = synth.molecule_batch(num_graphs=64), ["H", "C", "O"]
    else:
        d = Golden(case).data()
        elems = {"alloy108": ["Al", "Ni", "Cu"], "mol16": ["H", "C", "O"], "c1_si64": ["Si"]}[case]
    zl = [atomic_numbers[e] for e in elems]
    d = d.to(dev)
    args = (d.atomic_number, d.edge_index, zl, d.get("edge_shift"), d.batch)
    from hermnet_amd import switches
    monkeypatch.setattr(switches, "native_relations", False)
    ref = RelationalGraph.build_triadic(*args)
    monkeypatch.setattr(switches, "native_relations", True)
    for _ in range(2):
        g = RelationalGraph.build_triadic(*args)
        assert (g.N, g.E, g.T, g.num_src, g.block, g.triadic_pairs, g.num_graphs) == \
               (ref.N, ref.E, ref.T, ref.num_src, ref.block, ref.triadic_pairs, ref.num_graphs)
        assert g.type_rowptr_host == ref.type_rowptr_host
        for k in ["type_rowptr", "node_order", "row_of_node", "z_rows", "src_real", "row_real", "row_active", "res_row",
                  "csr_rowptr", "csr_src", "csr_perm", "src_id", "tgt_id", "shift", "csc_rowptr", "csc_tgt", "csc_pos",
                  "src_ranges", "batch32"]:
            a, b = getattr(g, k), getattr(ref, k)
            assert (a is None) == (b is None), k
            if a is not None:
                assert a.shape == b.shape and torch.equal(a.long() if not a.is_floating_point() else a,
                                                          b.long() if not b.is_floating_point() else b), k


@pytest.mark.gpu
def test_single_element_htnet_is_hvnet_gpu():
    g = Golden("c1_si64")
    hv = g.model()
    ht = hn.HTNet(["Si"], **g.model_kw).eval()
    ht.load_state_dict({k.replace("mods.Si.", "mods.Si_Si-Si."): v for k, v in hv.state_dict().items()})
    for p in ht.parameters():
        p.requires_grad_(False)
    e, f = _ef(ht.to(_dev()), g.data().to(_dev()))
    assert rel_err(e.cpu(), g.energy) < 1e-5 and rel_err(f.cpu(), g.forces) < 1e-5      # the reference's own numbers


def _rotation(seed=0):
    q, _ = np.linalg.qr(np.random.RandomState(seed).normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return torch.from_numpy(q.astype(np.float32))


@pytest.mark.gpu
def test_invariances_and_extensivity():
    """Properties that hold for any weights: rotating the structure rotates the forces and leaves the energy;
    relabelling the atoms permutes the forces; two copies far apart give twice the energy; forces sum to zero."""
    dev = _dev()
    elems = ["H", "C", "O"]
    model, sd = _model(elems, dict(rc=5.0, num_layers=3, hidden_channels=128, num_rbf=64))
    model = model.to(dev)
    mol = synth.molecule_batch(num_graphs=1, nmin=24, nmax=24, seed=3)
    e0, f0 = _ef(model, hn.Data(**{k: v.clone() for k, v in mol}).to(dev))
    assert float(f0.sum(0).abs().max()) < 2e-5 * float(f0.abs().max()) * mol.pos.size(0)
    # rotation + translation
    R = _rotation(1)
    d = hn.Data(**{k: v.clone() for k, v in mol})
    d.pos = d.pos @ R.T + torch.tensor([1.5, -2.0, 0.7])
    e1, f1 = _ef(model, d.to(dev))
    assert rel_err(e1, e0) < 1e-5 and rel_err(f1.cpu(), f0.cpu() @ R.T) < 2e-5
    # permutation of the atom labels
    perm = torch.randperm(mol.pos.size(0), generator=torch.Generator().manual_seed(2))
    inv = torch.argsort(perm)
    d = hn.Data(pos=mol.pos[perm].clone(), atomic_number=mol.atomic_number[perm].clone(),
                edge_index=inv[mol.edge_index], batch=mol.batch.clone())
    e2, f2 = _ef(model, d.to(dev))
    assert rel_err(e2, e0) < 1e-5 and rel_err(f2.cpu(), f0.cpu()[perm]) < 2e-5
    # extensivity: two copies 12 A apart (diameter 6 A + cutoff 5 A: no edges between them), one graph.  The shifted
    # copy's coordinates carry fp32 rounding of the larger numbers, hence the wider force tolerance for it.
    n = mol.pos.size(0)
    d = hn.Data(pos=torch.cat([mol.pos, mol.pos + 12.0]), atomic_number=torch.cat([mol.atomic_number] * 2),
                edge_index=torch.cat([mol.edge_index, mol.edge_index + n], 1), batch=torch.zeros(2 * n, dtype=torch.long))
    e3, f3 = _ef(model, d.to(dev))
    assert rel_err(e3, 2 * e0) < 1e-5 and rel_err(f3[:n].cpu(), f0.cpu()) < 2e-5 and rel_err(f3[n:].cpu(), f0.cpu()) < 5e-5


@pytest.mark.gpu
def test_config3_10k_atoms_htnet_properties_and_sampled_oracle():
    """BASELINE configs[2] at full size: 10,000-atom 3-element cell, 18 triadic relations, H = 128, R = 128, 5 layers.
    The oracle at this size needs ~3x HVNet's memory/time on the CPU, so the full-size check is through properties
    (finite, forces sum to zero, run-to-run bit-identical, translation invariance), plus the oracle on a 500-atom
    cell of the same alloy with the same weights."""
    dev = _dev()
    elems = ["Al", "Ni", "Cu"]
    kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
    model, sd = _model(elems, kw, seed=10)
    model = model.to(dev)
    d = synth.fcc_alloy(device=dev)
    e, f = _ef(model, d)
    assert torch.isfinite(e).all() and torch.isfinite(f).all()
    assert float(f.sum(0).abs().max()) < 1e-3 * float(f.abs().max()) * math.sqrt(d.pos.size(0))
    d.pos.grad = None
    e2, f2 = _ef(model, d)
    assert torch.equal(e, e2) and torch.equal(f, f2)
    small = synth.fcc_alloy(reps=(5, 5, 5))
    e_ref, f_ref = orc.htnet_energy_and_forces(sd, elems, small, **kw)
    es, fs = _ef(model, small.to(dev))
    assert rel_err(es.cpu(), e_ref) < 1e-5 and rel_err(fs.cpu(), f_ref) < 1e-5
