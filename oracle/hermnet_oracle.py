"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU) restatement of the reference's HVNet energy path, used only
as the checker: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; nothing under `hermnet_amd/` does.

Parity status: PINNED against the reference's own `HermNet/{hermnet,rmnet,utils}.py`
executed in the build container (fixtures in `tests/golden/`, generator
`tests/golden/gen_golden.py`).  The reference's third-party primitives
(torch_geometric.MessagePassing / GaussianSmearing / Data, torch_scatter.scatter,
ase.data.atomic_numbers -- all un-vendored and un-pinned, `requirements.txt:1-6`)
are restated here from their published semantics; the reference ships no tests,
so those boundaries are unpinned by the reference itself.

The functions take a `state_dict` with the reference's key layout
 and plain tensors; they work in whatever dtype the
inputs/weights are (float32 for parity, float64 for finite-difference checks).

Two execution modes of the heterogeneous layer:
  * mode="faithful"   -- same operation sequence as the reference, including the
                         per-node O(N*E) edge-mask loop of `utils.py:11-24` and
                         the run-on-all-N-then-keep-rows-of-type-t structure of
                         `hermnet.py:51-61`.  This is "the reference CPU path"
                         timed as `cpu_baseline` (kind "port").
  * mode="vectorised" -- one boolean mask per relation, update restricted to the
                         rows that survive; same arithmetic, no O(N*E) loop.
"""
import math

import torch
import torch.nn.functional as F

# ase.data.chemical_symbols (119 entries incl. 'X'); `hermnet.py:95` sizes the embedding with it.
CHEMICAL_SYMBOLS = [
    'X', 'H', 'He', 'Li', 'Be', 'B', 'C', 'N', 'O', 'F', 'Ne', 'Na', 'Mg', 'Al', 'Si', 'P', 'S', 'Cl', 'Ar',
    'K', 'Ca', 'Sc', 'Ti', 'V', 'Cr', 'Mn', 'Fe', 'Co', 'Ni', 'Cu', 'Zn', 'Ga', 'Ge', 'As', 'Se', 'Br', 'Kr',
    'Rb', 'Sr', 'Y', 'Zr', 'Nb', 'Mo', 'Tc', 'Ru', 'Rh', 'Pd', 'Ag', 'Cd', 'In', 'Sn', 'Sb', 'Te', 'I', 'Xe',
    'Cs', 'Ba', 'La', 'Ce', 'Pr', 'Nd', 'Pm', 'Sm', 'Eu', 'Gd', 'Tb', 'Dy', 'Ho', 'Er', 'Tm', 'Yb', 'Lu',
    'Hf', 'Ta', 'W', 'Re', 'Os', 'Ir', 'Pt', 'Au', 'Hg', 'Tl', 'Pb', 'Bi', 'Po', 'At', 'Rn',
    'Fr', 'Ra', 'Ac', 'Th', 'Pa', 'U', 'Np', 'Pu', 'Am', 'Cm', 'Bk', 'Cf', 'Es', 'Fm', 'Md', 'No', 'Lr',
    'Rf', 'Db', 'Sg', 'Bh', 'Hs', 'Mt', 'Ds', 'Rg', 'Cn', 'Nh', 'Fl', 'Mc', 'Lv', 'Ts', 'Og']
ATOMIC_NUMBERS = {s: z for z, s in enumerate(CHEMICAL_SYMBOLS)}


def scaled_silu(x):
    """`rmnet.py:110-117`: silu(x) * (1/0.6)."""
    return F.silu(x) * (1.0 / 0.6)


def edge_geometry(pos, edge_index, edge_shift=None, cell=None, batch=None):
    """`hermnet.py:133-152` with `torch.where` instead of the in-place write at :147
    (the in-place form breaks autograd; values are identical)."""
    j, i = edge_index[0], edge_index[1]
    dvec = pos[j] - pos[i]
    if cell is not None and edge_shift is not None:
        c = cell.reshape(-1, 3, 3)
        dvec = dvec + torch.einsum('ni,nij->nj', edge_shift.to(pos.dtype), c[batch[j]].to(pos.dtype))
    dist = dvec.norm(dim=-1)
    near0 = torch.isclose(dist, torch.zeros((), dtype=dist.dtype), atol=1e-6)
    dist = torch.where(near0, torch.full_like(dist, 1.0e-6), dist)
    return dist, dvec / dist[:, None]


def envelope(u, spec):
    """`rmnet.py:175-208`."""
    name = spec["name"].lower()
    if name == "polynomial":
        p = spec["exponent"]
        a = -(p + 1) * (p + 2) / 2
        b = p * (p + 2)
        c = -p * (p + 1) / 2
        val = 1 + a * u ** p + b * u ** (p + 1) + c * u ** (p + 2)
    elif name == "exponential":
        val = torch.exp(-(u ** 2) / ((1 - u) * (1 + u)))
    else:
        raise ValueError(name)
    return torch.where(u < 1, val, torch.zeros_like(u))


def radial_basis(dist, sd, rc, num_rbf, rbf_spec, env_spec):
    """`rmnet.py:168-172`: env(d/rc)[:,None] * rbf(d/rc) -> [E, R]."""
    u = dist * (1.0 / rc)
    env = envelope(u, env_spec)
    name = rbf_spec["name"].lower()
    if name == "gaussian":
        # PyG GaussianSmearing(start=0, stop=1, num_gaussians=R): buffer `offset`, python-float coeff
        offset = sd["radial_basis.rbf.offset"].to(dist.dtype)
        coeff = -0.5 / float(offset[1] - offset[0]) ** 2
        rbf = torch.exp(coeff * (u.view(-1, 1) - offset.view(1, -1)) ** 2)
    elif name == "spherical_bessel":
        freq = sd["radial_basis.rbf.frequencies"].to(dist.dtype)
        rbf = math.sqrt(2.0 / rc ** 3) / u[:, None] * torch.sin(freq * u[:, None])
    elif name == "bernstein":
        from scipy.special import binom
        import numpy as np
        pref = torch.tensor(binom(num_rbf - 1, np.arange(num_rbf)), dtype=torch.float).to(dist.dtype)
        gamma = F.softplus(sd["radial_basis.rbf.pregamma"].to(dist.dtype))
        e1 = torch.arange(num_rbf)[None, :]
        e2 = num_rbf - 1 - e1
        ed = torch.exp(-gamma * u)[:, None]
        rbf = pref * (ed ** e1) * ((1 - ed) ** e2)
    else:
        raise ValueError(name)
    return env[:, None] * rbf


def painn_message(p, x, vec, src, tgt, edge_embed, edge_vec, H, n_out):
    """`rmnet.py:51-73`: node MLP on LayerNorm(x), Linear on the edge basis, gather by
    source, PaiNN scalar/vector message, sum by target."""
    xn = F.layer_norm(x, (H,), p["x_layernorm.weight"], p["x_layernorm.bias"], 1e-5)
    xh = F.linear(scaled_silu(F.linear(xn, p["x_proj.0.weight"], p["x_proj.0.bias"])),
                  p["x_proj.2.weight"], p["x_proj.2.bias"])
    rbfh = F.linear(edge_embed, p["rbf_proj.weight"], p["rbf_proj.bias"])
    xh_j = xh.index_select(0, src)
    vec_j = vec.index_select(0, src)
    m = xh_j * rbfh
    s, a, b = torch.split(m, H, dim=-1)
    a = a * (1.0 / math.sqrt(3.0))
    mv = vec_j * a.unsqueeze(1) + b.unsqueeze(1) * edge_vec.unsqueeze(2)
    mv = mv * (1.0 / math.sqrt(H))
    dx = torch.zeros(n_out, H, dtype=x.dtype).index_add_(0, tgt, s)
    dvec = torch.zeros(n_out, 3, H, dtype=x.dtype).index_add_(0, tgt, mv)
    return dx, dvec


def painn_update(p, x, vec, H):
    """`rmnet.py:94-107`."""
    vp = F.linear(vec, p["vec_proj.weight"])
    v1, v2 = torch.split(vp, H, dim=-1)
    vdot = (v1 * v2).sum(dim=1) * (1.0 / math.sqrt(H))
    nrm = torch.sqrt(torch.sum(v2 ** 2, dim=-2) + 1e-8)
    h = F.linear(scaled_silu(F.linear(torch.cat([x, nrm], dim=-1), p["xvec_proj.0.weight"], p["xvec_proj.0.bias"])),
                 p["xvec_proj.2.weight"], p["xvec_proj.2.bias"])
    q1, q2, q3 = torch.split(h, H, dim=-1)
    dx = (q1 + q2 * vdot) * (1.0 / math.sqrt(2.0))
    dvec = q3.unsqueeze(1) * v1
    return dx, dvec


def _module_params(sd, layer, elem):
    pre = "hermconvs.%d.mods.%s." % (layer, elem)
    out = {}
    for k, v in sd.items():
        if k.startswith(pre + "message_layer."):
            out[k[len(pre + "message_layer."):]] = v
        elif k.startswith(pre + "update_layer."):
            out[k[len(pre + "update_layer."):]] = v
    return out


def painn_module(p, x, vec, src, tgt, edge_embed, edge_vec, H):
    """`rmnet.py:21-32` on all N rows (returns x, vec)."""
    n = x.size(0)
    dx, dvec = painn_message(p, x, vec, src, tgt, edge_embed, edge_vec, H, n)
    x = (x + dx) * (1.0 / math.sqrt(2.0))
    vec = vec + dvec
    dx, dvec = painn_update(p, x, vec, H)
    return x + dx, vec + dvec


def hetero_layer(sd, layer, elems, x, vec, z, edge_index, edge_embed, edge_vec, H, mode):
    """`hermnet.py:37-65` (+ `utils.py:11-24`)."""
    x_out = torch.zeros_like(x)
    v_out = torch.zeros_like(vec)
    for el in elems:
        zt = ATOMIC_NUMBERS[el]
        nid = torch.where(z == zt)[0]
        if mode == "faithful":
            if nid.numel() == 0:
                # the reference crashes here (`torch.cat([])`); the build treats it as a no-op
                continue
            emask = torch.cat([torch.where(edge_index[1] == n)[0] for n in nid])
        else:
            emask = torch.where((z == zt)[edge_index[1]])[0]
        if emask.numel() == 0:
            continue
        p = _module_params(sd, layer, el)
        src, tgt = edge_index[0][emask], edge_index[1][emask]
        ee, ev = edge_embed[emask], edge_vec[emask]
        if mode == "faithful":
            xo, vo = painn_module(p, x, vec, src, tgt, ee, ev, H)
            v_out = v_out.index_add(0, nid, vo[nid])
            x_out = x_out.index_add(0, nid, xo[nid])
        else:
            n = x.size(0)
            dx, dvec = painn_message(p, x, vec, src, tgt, ee, ev, H, n)
            x1 = (x[nid] + dx[nid]) * (1.0 / math.sqrt(2.0))
            v1 = vec[nid] + dvec[nid]
            dx2, dv2 = painn_update(p, x1, v1, H)
            x_out = x_out.index_copy(0, nid, x1 + dx2)
            v_out = v_out.index_copy(0, nid, v1 + dv2)
    return x_out, v_out


def hvnet_energy(sd, elems, pos, z, edge_index, batch, edge_shift=None, cell=None, *, rc=5.0,
                 intensive=False, num_layers=5, hidden_channels=128, num_rbf=128,
                 rbf=None, envelope_spec=None, mode="vectorised", return_intermediates=False):
    """`hermnet.py:118-131`: energy per graph [num_graphs]."""
    rbf = rbf or {"name": "gaussian"}
    envelope_spec = envelope_spec or {"name": "polynomial", "exponent": 5}
    if isinstance(elems, str):
        elems = [elems]  # intent of `Union[str, List[str]]` (`hermnet.py:84`): one symbol
    H = hidden_channels
    dist, evec = edge_geometry(pos, edge_index, edge_shift, cell, batch)
    eemb = radial_basis(dist, sd, rc, num_rbf, rbf, envelope_spec)
    x = sd["embed.weight"][z.long()]
    vec = torch.zeros(x.size(0), 3, H, dtype=x.dtype)
    inter = {"edge_dist": dist, "edge_vec": evec, "edge_embed": eemb, "x": [], "vec": []}
    for l in range(num_layers):
        x, vec = hetero_layer(sd, l, elems, x, vec, z, edge_index, eemb, evec, H, mode)
        if return_intermediates:
            inter["x"].append(x)
            inter["vec"].append(vec)
    e_atom = F.linear(scaled_silu(F.linear(x, sd["out_energy.0.weight"], sd["out_energy.0.bias"])),
                      sd["out_energy.2.weight"], sd["out_energy.2.bias"]).squeeze(1)
    ng = int(batch.max()) + 1 if batch.numel() else 0
    energy = torch.zeros(ng, dtype=e_atom.dtype).index_add_(0, batch, e_atom)
    if intensive:
        cnt = torch.zeros(ng, dtype=e_atom.dtype).index_add_(0, batch, torch.ones_like(e_atom)).clamp_(min=1)
        energy = energy / cnt
    if return_intermediates:
        return energy, inter
    return energy


# ---------------------------------------------------------------------------------------------
# HTNet (Heterogeneous Triadic Network) -- PARITY UNPINNED.
# The reference names the model (`README.md:27`) and draws its subgraphs (`figs/subgraph.svg`, panel (c):
# "A->A<-A:", "B->A<-C:"), but its class is a stub that raises NotImplementedError
# (`HermNet/hermnet.py:155-157`), so there is nothing to pin against.  What follows restates the BUILD-DEFINED
# specification in DESIGN.md ("HTNet"): it is the checker of the HIP path for BASELINE configs[2], not a
# statement about the reference's numerics.
# ---------------------------------------------------------------------------------------------
def triadic_relations(elems):
    """[(key, centre, (p, q))]: one relation per centre element and UNORDERED pair of neighbour elements, in
    module order: centres in `elems` order, pairs (p <= q) in `elems` order.  T * T(T+1)/2 relations (18 for T = 3)."""
    out = []
    for c in elems:
        for i, p in enumerate(elems):
            for q in elems[i:]:
                out.append(("%s_%s-%s" % (c, p, q), c, (p, q)))
    return out


def triadic_layer(sd, layer, elems, x, vec, z, edge_index, edge_embed, edge_vec, H):
    """One heterogeneous triadic layer: the loop of `hermnet.py:51-61` over triadic relations instead of
    elements.  Relation (c; p, q) owns a PaiNNModule; its subgraph = the edges j -> i with element(i) = c and
    element(j) in {p, q} (`in_subgraph` restricted by source element, figs/subgraph.svg (c)); relations without an
    edge are skipped (`hermnet.py:56-57`); the results of a centre's T(T+1)/2 relations are AVERAGED (the
    reference accumulates with `+=`, `hermnet.py:60-61`; the fixed 1/P keeps activations at the scale of HVNet and
    makes HTNet with one element identical to HVNet)."""
    T = len(elems)
    P = T * (T + 1) // 2
    x_out = torch.zeros_like(x)
    v_out = torch.zeros_like(vec)
    src_all, tgt_all = edge_index[0], edge_index[1]
    for key, c, (p, q) in triadic_relations(elems):
        is_c = z == ATOMIC_NUMBERS[c]
        from_pq = (z == ATOMIC_NUMBERS[p]) | (z == ATOMIC_NUMBERS[q])
        emask = torch.where(is_c[tgt_all] & from_pq[src_all])[0]
        if emask.numel() == 0:
            continue
        nid = torch.where(is_c)[0]
        prm = _module_params(sd, layer, key)
        dx, dvec = painn_message(prm, x, vec, src_all[emask], tgt_all[emask], edge_embed[emask], edge_vec[emask], H,
                                 x.size(0))
        x1 = (x[nid] + dx[nid]) * (1.0 / math.sqrt(2.0))
        v1 = vec[nid] + dvec[nid]
        dx2, dv2 = painn_update(prm, x1, v1, H)
        x_out = x_out.index_add(0, nid, (x1 + dx2) * (1.0 / P))
        v_out = v_out.index_add(0, nid, (v1 + dv2) * (1.0 / P))
    return x_out, v_out


def htnet_energy(sd, elems, pos, z, edge_index, batch, edge_shift=None, cell=None, *, rc=5.0, intensive=False,
                 num_layers=5, hidden_channels=128, num_rbf=128, rbf=None, envelope_spec=None):
    """Energy per graph of the build-defined HTNet: `hvnet_energy` with triadic layers."""
    rbf = rbf or {"name": "gaussian"}
    envelope_spec = envelope_spec or {"name": "polynomial", "exponent": 5}
    if isinstance(elems, str):
        elems = [elems]
    H = hidden_channels
    dist, evec = edge_geometry(pos, edge_index, edge_shift, cell, batch)
    eemb = radial_basis(dist, sd, rc, num_rbf, rbf, envelope_spec)
    x = sd["embed.weight"][z.long()]
    vec = torch.zeros(x.size(0), 3, H, dtype=x.dtype)
    for l in range(num_layers):
        x, vec = triadic_layer(sd, l, elems, x, vec, z, edge_index, eemb, evec, H)
    e_atom = F.linear(scaled_silu(F.linear(x, sd["out_energy.0.weight"], sd["out_energy.0.bias"])),
                      sd["out_energy.2.weight"], sd["out_energy.2.bias"]).squeeze(1)
    ng = int(batch.max()) + 1 if batch.numel() else 0
    energy = torch.zeros(ng, dtype=e_atom.dtype).index_add_(0, batch, e_atom)
    if intensive:
        cnt = torch.zeros(ng, dtype=e_atom.dtype).index_add_(0, batch, torch.ones_like(e_atom)).clamp_(min=1)
        energy = energy / cnt
    return energy


def htnet_energy_and_forces(sd, elems, data, **kw):
    pos = data.pos.detach().clone().requires_grad_(True)
    e = htnet_energy(sd, elems, pos, data.atomic_number, data.edge_index, data.batch,
                     data.get("edge_shift"), data.get("cell"), **kw)
    if not e.requires_grad:
        return e.detach(), torch.zeros_like(pos)
    g = torch.autograd.grad(e.sum(), pos, allow_unused=True)[0]
    return e.detach(), (torch.zeros_like(pos) if g is None else -g)


def energy_and_forces(sd, elems, data, **kw):
    """Energy [num_graphs] and forces [N,3] = -d(sum E)/d pos (`calculator.py:75-83`)."""
    pos = data.pos.detach().clone().requires_grad_(True)
    e = hvnet_energy(sd, elems, pos, data.atomic_number, data.edge_index, data.batch,
                     data.get("edge_shift"), data.get("cell"), **kw)
    if not e.requires_grad:          # no edge at all: the energy does not depend on the coordinates
        return e.detach(), torch.zeros_like(pos)
    g = torch.autograd.grad(e.sum(), pos, allow_unused=True)[0]
    f = torch.zeros_like(pos) if g is None else -g
    return e.detach(), f


def htnet_training_loss_and_grads(sd, elems, data, y, forces, gamma=0.8, **kw):
    """`training_loss_and_grads` for the build-defined HTNet (same loss, `example/dist_train.py:86-99`): the checker of
    HTNet's train() mode.  Parity unpinned, like everything about HTNet: the reference's class is a stub."""
    buffers = ("radial_basis.rbf.offset",)
    p = {k: (v.detach().clone().requires_grad_(True) if (k not in buffers and v.is_floating_point()) else v)
         for k, v in sd.items()}
    pos = data.pos.detach().clone().requires_grad_(True)
    e = htnet_energy(p, elems, pos, data.atomic_number, data.edge_index, data.batch,
                     data.get("edge_shift"), data.get("cell"), **kw)
    e_loss = F.mse_loss(e, y)
    f = -torch.autograd.grad(e.sum(), pos, create_graph=True)[0]
    f_loss = F.mse_loss(f, forces)
    loss = (1 - gamma) * e_loss + gamma * f_loss
    keys = [k for k, v in p.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [p[k] for k in keys], allow_unused=True)
    return loss.detach(), e_loss.detach(), f_loss.detach(), dict(zip(keys, grads))


def training_loss_and_grads(sd, elems, data, y, forces, gamma=0.8, **kw):
    """One optimisation step's loss and parameter gradients, `example/dist_train.py:86-99`:
    e_loss = MSE(E, y), F = -d(sum E)/d pos with create_graph=True, f_loss = MSE(F, forces),
    loss = (1-gamma) e_loss + gamma f_loss, loss.backward().  Returns (loss, e_loss, f_loss,
    {state_dict key: gradient or None}); buffers (e.g. `radial_basis.rbf.offset`) are not differentiated."""
    buffers = ("radial_basis.rbf.offset",)
    p = {k: (v.detach().clone().requires_grad_(True) if (k not in buffers and v.is_floating_point()) else v)
         for k, v in sd.items()}
    pos = data.pos.detach().clone().requires_grad_(True)
    e = hvnet_energy(p, elems, pos, data.atomic_number, data.edge_index, data.batch,
                     data.get("edge_shift"), data.get("cell"), **kw)
    e_loss = F.mse_loss(e, y)
    f = -torch.autograd.grad(e.sum(), pos, create_graph=True)[0]
    f_loss = F.mse_loss(f, forces)
    loss = (1 - gamma) * e_loss + gamma * f_loss
    keys = [k for k, v in p.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [p[k] for k in keys], allow_unused=True)
    return loss.detach(), e_loss.detach(), f_loss.detach(), dict(zip(keys, grads))
