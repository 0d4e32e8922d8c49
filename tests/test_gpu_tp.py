"""-m gpu: the tensor-parallel decode step ON THE DEVICE (proxy_inference_engine_amd/tp.py: TPLlama), world_size 2.

The one-GPU box has one card, so both ranks share cuda:0 and the collectives run over gloo (the code path is
torch.distributed's; on an 8-GPU node the backend is "nccl" = RCCL over xGMI).  Each rank shards the same int4 checkpoint,
decodes a short sequence greedily with one all-reduce per row-parallel Linear, and rank 0 compares every step with the
oracle on the UNSHARDED checkpoint: same tokens where the margin allows, hidden state within the end-to-end tolerance."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pie_oracle as po
from tests._util import EPS, assert_vec_close, codes_dev, to_dev

pytestmark = pytest.mark.gpu
DT = "bfloat16"
CFG = {"model_type": "llama", "hidden_size": 512, "num_hidden_layers": 2, "intermediate_size": 1024,
       "num_attention_heads": 8, "num_key_value_heads": 4, "rms_norm_eps": 1e-5, "vocab_size": 1024,
       "rope_theta": 10000.0, "max_position_embeddings": 2048, "tie_word_embeddings": False,
       "quantization": {"group_size": 64, "bits": 4}}
PROMPT_SEED, STEPS = 3, 10


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from proxy_inference_engine_amd.tp import TPGroup, TPLlama
        torch.cuda.set_device(0)
        w = po.synth_checkpoint(CFG, seed=71, dtype=DT, lm_head_gain=4.0)
        dev_w = {k: (codes_dev(v) if v.dtype == np.uint32 else to_dev(v, DT)) for k, v in w.items()}
        model = TPLlama(CFG, dev_w, TPGroup())
        prompt = np.random.default_rng(PROMPT_SEED).integers(0, CFG["vocab_size"], 6)
        out = []
        tok = None
        for t in prompt:                                               # prompt by iterated steps, then greedy decode
            tok, lse, hid = model.step(int(t))
        for _ in range(STEPS):
            out.append((tok, lse, hid.float().cpu().numpy().copy()))
            tok, lse, hid = model.step(tok)
        if rank == 0:
            ret["tokens"] = [o[0] for o in out]
            ret["lse"] = [o[1] for o in out]
            ret["hidden"] = [o[2] for o in out]
        ret[f"rank{rank}_tokens"] = [o[0] for o in out]
    finally:
        dist.destroy_process_group()


def test_tp2_decode_matches_unsharded_oracle():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["rank0_tokens"] == ret["rank1_tokens"]                 # both ranks decode the same sequence
    w = po.synth_checkpoint(CFG, seed=71, dtype=DT, lm_head_gain=4.0)
    orc = po.OracleLlama(CFG, w, DT)
    prompt = np.random.default_rng(PROMPT_SEED).integers(0, CFG["vocab_size"], 6)
    ocache = [po.OracleKVCache() for _ in orc.layers]
    po.set_qmm_min_rows(0)
    try:
        logits, hid = orc.forward(prompt, ocache, want_hidden=True)
        logits, hid = logits[-1], hid[-1]
        checked = 0
        for i in range(STEPS):
            assert_vec_close(ret["hidden"][i][0], hid, DT, what=f"TP hidden step {i}")
            otok, olp = po.logprobs_argmax(logits)
            lse_ref = float(np.log(np.exp(logits.astype(np.float64) - logits.max()).sum()) + logits.max())
            assert abs(ret["lse"][i] - lse_ref) <= 4 * EPS[DT] * np.abs(logits).max()
            top2 = np.sort(olp)[-2:]
            if top2[1] - top2[0] > 2 * 4 * EPS[DT] * np.abs(logits).max():
                assert ret["tokens"][i] == otok, f"step {i}"
                checked += 1
            logits, hid = orc.forward(np.array([ret["tokens"][i]]), ocache, want_hidden=True)   # teacher-forced with the TP tokens
            logits, hid = logits[0], hid[0]
        assert checked >= 3
    finally:
        po.set_qmm_min_rows(6)


# ---------------------------------------------------------------- the FUSED tensor-parallel step (product path): tp.fused_shard + HipComm
def _fused_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = None
    try:
        from proxy_inference_engine_amd.tp import HipComm, TPGroup, TPLlama, fused_shard
        torch.cuda.set_device(0)
        w = po.synth_checkpoint(CFG, seed=71, dtype=DT, lm_head_gain=4.0)
        dev_w = {k: (codes_dev(v) if v.dtype == np.uint32 else to_dev(v, DT)) for k, v in w.items()}
        comm = HipComm(8192)
        # the collective alone: rank-order fp32 sum, in place, stream-ordered; one workgroup (n = 512), several (8192), an odd length
        for n in (CFG["hidden_size"], 8192, 4097):
            both = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r), dtype=torch.float32) for r in range(world)]
            for rep in range(5):  # epochs alternate the two receive areas
                buf = (both[rank] * (rep + 1)).cuda()
                comm.all_reduce(buf)
                want = both[0] * (rep + 1)
                for r in range(1, world):
                    want = want + both[r] * (rep + 1)
                assert torch.equal(buf.cpu(), want), f"all-reduce n={n} rep {rep}"
        functional = TPLlama(CFG, dev_w, TPGroup())
        fused = fused_shard(CFG, dev_w, comm)
        cache = fused.make_cache()
        prompt = np.random.default_rng(PROMPT_SEED).integers(0, CFG["vocab_size"], 6)
        tok = None
        for t in prompt:
            tok, lse, hid = functional.step(int(t))
        ftok, flp, flog = fused.step(torch.tensor(prompt, dtype=torch.int32, device="cuda"), cache)  # prompt through the step kernels
        f_tokens, t_tokens, hid_err, lse_err = [], [], [], []
        V_loc = CFG["vocab_size"] // world
        for i in range(STEPS):
            f_tokens.append(int(ftok.item()))
            t_tokens.append(int(tok))
            hid_err.append(float((fused.hidden.float() - hid[0].float()).abs().max().item() / max(float(hid.float().abs().max().item()), 1e-6)))
            # logprobs = logits - lse on this rank's vocabulary rows
            f_lse = (flog.float() - flp)[:V_loc]
            lse_err.append(float((f_lse - lse).abs().max().item()))
            assert flp.numel() == V_loc and flog.numel() == V_loc
            tok, lse, hid = functional.step(tok)
            # eager for the first steps, then the captured hipGraph (the collectives replay inside it)
            ftok, flp, flog = fused.step(None, cache, graph=i >= 2)
        ret[f"fused{rank}"] = f_tokens
        ret[f"func{rank}"] = t_tokens
        ret[f"hid_err{rank}"] = max(hid_err)
        ret[f"lse_err{rank}"] = max(lse_err)
        ret[f"status{rank}"] = comm.status()
        del fused
    finally:
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


def test_tp2_fused_step_equals_functional_tp():
    """Two ranks on one card: the fused step (decoder launches + pie_comm all-reduce, eager and as a hipGraph) produces the same
    greedy tokens as the functional TPLlama reference on both ranks, hidden state and log-sum-exp within the 16-bit tolerance."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_fused_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["status0"] == 0 and ret["status1"] == 0
    assert ret["fused0"] == ret["fused1"] == ret["func0"] == ret["func1"], (ret["fused0"], ret["func0"])
    assert ret["hid_err0"] <= 4 * EPS[DT] and ret["hid_err1"] <= 4 * EPS[DT]
    assert ret["lse_err0"] <= 0.05 and ret["lse_err1"] <= 0.05


@pytest.mark.parametrize("backend", ["ipc", "rccl"])
def test_one_rank_tensor_parallel_step_equals_the_unsharded_decoder(backend):
    """The decoder's tensor-parallel code path on ONE rank (tp_world = 1), both communicator backends: o_proj / down_proj through the
    collective (ipc: EPI_TP_PUSH granules to the rank's own area + the pull launch; rccl: fp32 partial + ncclAllReduce + the rounding launch),
    the vocabulary-parallel tail through its exchange (granules / ncclAllGather).  With one rank every sum has one addend, so logits, greedy
    tokens and the hidden state must equal the unsharded decoder's BIT FOR BIT, eager and as a captured hipGraph (RCCL captured with it).
    RCCL refuses two ranks on one device, so this is the only test of its backend a one-GPU box can run; the two-rank tests above run the
    one-shot backend."""
    from proxy_inference_engine_amd.models.llama import Model, ModelArgs
    from proxy_inference_engine_amd.tp import HipComm
    from tests._util import to_bits
    w = po.synth_checkpoint(CFG, seed=72, dtype=DT, lm_head_gain=4.0)
    dev_w = {k: (codes_dev(v) if v.dtype == np.uint32 else to_dev(v, DT)) for k, v in w.items()}
    plain = Model(ModelArgs(**CFG), dev_w)
    comm = HipComm(CFG["hidden_size"], backend=backend)
    assert comm.world == 1
    try:
        tp = Model(ModelArgs(**CFG), dev_w, tp=comm)
        prompt = torch.tensor(np.random.default_rng(5).integers(0, CFG["vocab_size"], 5), dtype=torch.int32, device="cuda")  # below 6 rows: both decoders iterate decode steps (one regime)
        ca, cb = plain.make_cache(), tp.make_cache()
        ta, la, ga = plain.step(prompt, ca)
        tb, lb, gb = tp.step(prompt, cb)
        for i in range(8):
            assert int(ta.item()) == int(tb.item()), f"step {i}"
            assert np.array_equal(to_bits(ga), to_bits(gb)), f"step {i}: logits"
            assert np.array_equal(to_bits(plain.hidden), to_bits(tp.hidden)), f"step {i}: hidden state"
            assert float((la - lb).abs().max().item()) <= 1e-5, f"step {i}: log-probabilities"   # the log-sum-exp is summed over another partition
            ta, la, ga = plain.step(None, ca, graph=i >= 2)
            tb, lb, gb = tp.step(None, cb, graph=i >= 2)
        assert comm.status() == 0
        del tp
    finally:
        comm.close()
