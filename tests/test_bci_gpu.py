"""GPU parity of the BCI coupler (llm_bci_amd/bci.py: HIP encoder -> HIP projector GEMMs -> HIP splice) against the fixture
produced by the reference's BCI.prepare_embeds (tests/golden/make_golden.py --bci)."""
import json
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from test_oracle_golden import load

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _StubLLM(nn.Module):
    """just enough of a HF causal LM for prepare_embeds: an embedding table, .config, .dtype"""

    def __init__(self, table):
        super().__init__()
        self.embed = nn.Embedding.from_pretrained(torch.from_numpy(table).clone(), freeze=False)
        self.config = types.SimpleNamespace(hidden_size=table.shape[1], vocab_size=table.shape[0])

    def get_input_embeddings(self):
        return self.embed

    @property
    def dtype(self):
        return self.embed.weight.dtype


def _build(fx, dtype):
    from llm_bci_amd.bci import BCI
    cfg = json.loads(str(fx["config_json"]))
    m = BCI(cfg, llm=_StubLLM(fx["embed_table"]), method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype=dtype)
    m.llm.float()
    m.ndt1.load_state_dict({k[len("w:ndt1."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w:ndt1.")})
    m.projector.load_state_dict({k[len("w:projector."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w:projector.")})
    return m.to(DEV)


def _inputs(fx):
    d = lambda k: torch.from_numpy(fx[k]).to(DEV)
    return (d("input_ids"), d("attention_mask"), d("input_split"), d("spikes"), d("spikes_mask"), d("spikes_timestamp"),
            d("spikes_lengths"), None, None, d("targets"))


def test_prepare_embeds_matches_reference_fp32():
    fx = load("g_bci")
    m = _build(fx, "fp32")
    m.eval()
    emb, mask, tg = m.prepare_embeds(*_inputs(fx))
    torch.cuda.synchronize()
    assert np.array_equal(mask.cpu().numpy(), fx["out_mask"])          # integer outputs: bit-exact
    assert np.array_equal(tg.cpu().numpy(), fx["out_targets"])
    np.testing.assert_allclose(emb.detach().float().cpu().numpy(), fx["out_embeds"], atol=1e-3)
    (emb.float() * torch.from_numpy(fx["R"]).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    for k in fx.files:
        if k.startswith("g:projector."):
            p = dict(m.projector.named_parameters())[k[len("g:projector."):]]
            ref = fx[k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-3 * max(1.0, np.abs(ref).max()), err_msg=k)
        if k.startswith("g:ndt1."):
            p = dict(m.ndt1.named_parameters())[k[len("g:ndt1."):]]
            ref = fx[k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-3 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_prepare_embeds_bf16_close_and_forward_runs():
    fx = load("g_bci")
    m = _build(fx, "bf16")
    m.eval()
    emb, mask, tg = m.prepare_embeds(*_inputs(fx))
    torch.cuda.synchronize()
    assert np.array_equal(mask.cpu().numpy(), fx["out_mask"]) and np.array_equal(tg.cpu().numpy(), fx["out_targets"])
    assert np.abs(emb.detach().float().cpu().numpy() - fx["out_embeds"]).max() < 0.06


def test_checkpoint_files(tmp_path):
    import os
    fx = load("g_bci")
    m = _build(fx, "fp32")
    m.llm.save_pretrained = lambda d: None     # the stub has no HF serialisation
    m.save_checkpoint(str(tmp_path))
    assert {"projector.bin", "projector_config.pth", "encoder.bin", "decoder.bin", "encoder_config.pth"} <= set(os.listdir(tmp_path))
    sd = torch.load(os.path.join(tmp_path, "projector.bin"))
    assert set(sd.keys()) == {"0.weight", "0.bias", "2.weight", "2.bias"}
