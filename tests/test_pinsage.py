"""PinSAGE path: sampler mirror laws on the CPU; on the GPU the device samplers against
the mirror bit for bit, the model against its torch-only twin, and training."""
import numpy as np
import pytest
import torch as t

from oracle import pinsage_ref as PR


def _graph(seed=0, U=120, I=60, E=900):
    from laplace_amd import synthetic as S
    from laplace_amd.data.dataset import AdjList
    ei = S.generate(S.SyntheticSpec(U, I, E, seed=seed, deg_min=1, deg_max=40, zipf_s=0.8))
    u, a = ei[0].numpy(), ei[1].numpy()
    users, items = AdjList.from_edges(u, a, U), AdjList.from_edges(a, u, I)
    return users, items, PR.Csr(users.ptr, users.idx), PR.Csr(items.ptr, items.idx), U, I


def test_mirror_walk_laws():
    users, items, ucsr, icsr, U, I = _graph()
    heads, tails, negs = PR.item_pairs(256, I, icsr, ucsr, seed=3, step=1)
    assert len(heads) == len(tails) == len(negs) <= 256 and (tails >= 0).all()
    for h, tl in zip(heads[:50], tails[:50]):  # tail is reachable by item -> user -> item
        assert any(tl in ucsr[int(u)] for u in icsr[int(h)])
    nb, wt = PR.pinsage_neighbors(np.arange(I), icsr, ucsr, walk_length=2, restart_prob=0.5, num_walks=10,
                                  num_neighbors=3, layer=0, seed=3, step=1)
    assert nb.shape == (I, 3) and (wt.sum(1) <= 20).all() and (wt[:, 0] >= wt[:, 1]).all() and (wt[:, 1] >= wt[:, 2]).all()
    assert ((nb >= 0) == (wt > 0)).all()
    # termination before the 2nd traversal with p = 0.5: mean visits per walk ~ 1.5 where walks cannot die
    full = np.array([wt_i.sum() for wt_i in PR.pinsage_neighbors(np.arange(I), icsr, ucsr, 2, 0.5, 10, 1000, 0, 5, 2)[1]])
    assert 13.0 < full.mean() < 17.0
    batch = PR.sample_from_item_pairs(heads, tails, negs, icsr, ucsr, 2, 2, 0.5, 10, 3, seed=3, step=1)
    b_in, b_out = batch["blocks"]
    assert b_out["n_dst"] == len(batch["seeds"]) and b_in["n_dst"] == len(b_out["src_ids"])
    assert np.array_equal(b_in["src_ids"][: b_in["n_dst"]], b_out["src_ids"])  # next layer's seeds = this layer's sources
    pairs = set(zip(heads.tolist(), tails.tolist())) | set(zip(heads.tolist(), negs.tolist()))
    for blk in batch["blocks"]:  # no frontier edge head -> tail survives (label leakage removal)
        src_g = blk["src_ids"][blk["edge_src"]]
        dst_g = blk["src_ids"][blk["edge_dst"]]
        assert not any((int(a), int(b)) in pairs for a, b in zip(src_g, dst_g))
        assert (blk["edge_dst"] < blk["n_dst"]).all() and (blk["weights"] >= 1).all()


@pytest.mark.gpu
def test_device_samplers_bit_exact_vs_mirror():
    from laplace_amd.pinsage.sampler import PinSAGESampler
    users, items, ucsr, icsr, U, I = _graph(seed=2, U=300, I=150, E=3000)
    for (L, p, W, T, layers) in [(2, 0.5, 10, 3, 2), (3, 0.25, 6, 5, 3), (1, 0.0, 4, 2, 1)]:
        smp = PinSAGESampler(users, items, U, I, batch_size=64, random_walk_length=L, random_walk_restart_prob=p,
                             num_random_walks=W, num_neighbors=T, num_layers=layers, seed=77)
        for step in (0, 5):
            h, tl, ng = smp.item_pairs(step)
            wh, wt_, wn = PR.item_pairs(64, I, icsr, ucsr, 77, step)
            assert np.array_equal(h.cpu().numpy(), wh) and np.array_equal(tl.cpu().numpy(), wt_) and np.array_equal(ng.cpu().numpy(), wn)
            got = smp.sample_batch(step)
            want = PR.sample_from_item_pairs(wh, wt_, wn, icsr, ucsr, layers, L, p, W, T, 77, step)
            assert np.array_equal(got["seeds"].cpu().numpy(), want["seeds"])
            for a, b in zip(got["pos"] + got["neg"], want["pos"] + want["neg"]):
                assert np.array_equal(a.cpu().numpy(), b)
            assert len(got["blocks"]) == layers
            for gb, wb in zip(got["blocks"], want["blocks"]):
                assert gb["n_dst"] == wb["n_dst"]
                for key in ("src_ids", "edge_src", "edge_dst"):
                    assert np.array_equal(gb[key].cpu().numpy(), wb[key]), key
                assert np.array_equal(gb["weights"].cpu().numpy(), wb["weights"])


@pytest.mark.gpu
def test_model_parity_and_training():
    from laplace_amd.pinsage.model import PinSAGEModel, train_epoch
    from laplace_amd.pinsage.sampler import PinSAGESampler
    users, items, ucsr, icsr, U, I = _graph(seed=4, U=400, I=200, E=5000)
    smp = PinSAGESampler(users, items, U, I, batch_size=64, seed=5)
    t.manual_seed(0)
    model = PinSAGEModel(I, 16, 2).to("cuda")
    ref = PR.PinSAGERef(I, 16, 2)
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    model.eval(); ref.eval()  # dropout off: identical arithmetic
    for step in range(3):
        b = smp.sample_batch(step)
        out = model(b["seeds"], b["pos"], b["neg"], b["blocks"])
        blocks_c = [{k: (v.cpu() if isinstance(v, t.Tensor) else v) for k, v in blk.items()} for blk in b["blocks"]]
        out_ref = ref(b["seeds"].cpu(), tuple(x.cpu() for x in b["pos"]), tuple(x.cpu() for x in b["neg"]), blocks_c)
        assert (out.detach().cpu() - out_ref.detach()).abs().max() <= 1e-4
        model.zero_grad(); ref.zero_grad()
        out.mean().backward(); out_ref.mean().backward()
        for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
            scale = float(pr.grad.abs().max()) + 1e-8
            assert float((p.grad.cpu() - pr.grad).abs().max()) <= 2e-4 * scale + 1e-7, n
    opt = t.optim.Adam(model.parameters(), lr=3e-3)
    first = train_epoch(model, opt, smp, 40)
    for _ in range(3):
        last = train_epoch(model, opt, smp, 40)
    assert np.isfinite(last).all() and np.mean(last) < np.mean(first)
