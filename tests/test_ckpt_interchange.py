"""Checkpoint interchange with the REFERENCE's own files. tests/golden/ckpt_<model>/ hold what the reference's
save_checkpoint wrote (tests/golden/make_golden.py --ckpt: ndt1.py:685-688, itransformer.py:403-407, patchtst.py:258-262, bci.py:250-257)
plus expected.npz (checksums of every state-dict tensor, eval outputs). Here the NATIVE classes read them: through `from_pt` (the
reference's warm start, ndt1.py:468-476,503-504) and through load_checkpoint; on the GPU the loaded model's eval output must equal
the reference's. The other direction - a native checkpoint loaded by the reference - needs the reference and is asserted inside
make_golden.py --ckpt; its record is tests/golden/ckpt_reverse_check.json."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _expected(name):
    return np.load(os.path.join(GOLD, name, "expected.npz"), allow_pickle=False)


def _check_state(sd, summary, skip=()):
    keys = [k for k in sd if not k.startswith(tuple(skip))] if skip else list(sd)
    assert sorted(keys) == sorted(summary), sorted(set(keys) ^ set(summary))
    for k in keys:
        shape, s, a = summary[k]
        v = sd[k].detach().double().cpu()
        assert list(v.shape) == shape, k
        assert abs(float(v.sum()) - s) <= 1e-6 * max(1.0, abs(a)) and abs(float(v.abs().sum()) - a) <= 1e-6 * max(1.0, abs(a)), k


def _ndt1(over=None, from_pt=None, dtype="fp32"):
    from llm_bci_amd.ndt1 import NDT1
    fx = _expected("ckpt_ndt1")
    over = json.loads(str(fx["config_json"]))
    if from_pt:
        over["encoder"]["from_pt"] = from_pt
    torch.manual_seed(123)   # (another initialisation than the checkpoint's)
    return NDT1(over, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype=dtype), fx


def test_ndt1_reads_the_reference_checkpoint_by_from_pt_and_by_load_checkpoint():
    d = os.path.join(GOLD, "ckpt_ndt1")
    m, fx = _ndt1(from_pt=d)
    _check_state(m.state_dict(), json.loads(str(fx["state_json"])))
    m2, _ = _ndt1()
    before = m2.state_dict()["encoder.layers.0.attn.query.weight"].clone()
    m2.load_checkpoint(d)
    assert not torch.equal(before, m2.state_dict()["encoder.layers.0.attn.query.weight"])
    _check_state(m2.state_dict(), json.loads(str(fx["state_json"])))


def test_itransformer_reads_the_reference_checkpoint():
    from llm_bci_amd.itransformer import iTransformer
    d = os.path.join(GOLD, "ckpt_itransformer")
    fx = _expected("ckpt_itransformer")
    over = json.loads(str(fx["config_json"]))
    o2 = json.loads(json.dumps(over)); o2["encoder"]["from_pt"] = d; o2.setdefault("decoder", {})["from_pt"] = d
    torch.manual_seed(123)
    m = iTransformer(o2, method_name="mlm", log_input=True, loss="poisson_nll", compute_dtype="fp32")
    _check_state(m.state_dict(), json.loads(str(fx["state_json"])), skip=("masker",))
    torch.manual_seed(5)
    m2 = iTransformer(over, method_name="mlm", log_input=True, loss="poisson_nll", compute_dtype="fp32")
    m2.load_checkpoint(d)
    _check_state(m2.state_dict(), json.loads(str(fx["state_json"])), skip=("masker",))


@pytest.mark.parametrize("dname", ["ckpt_itransformer_full", "ckpt_itransformer_uni"])
def test_itransformer_reads_reference_checkpoints_with_region_depth_tables_and_the_transformer_embedder(dname):
    """round 4: region + depth embedding tables (`_full`) and the UnivariateTransformer embedder + embed_proj (`_uni`) - every tensor of the
    reference-written files lands in the native state dict (make_golden.py --ckpt-itr)."""
    from llm_bci_amd.itransformer import iTransformer
    d = os.path.join(GOLD, dname)
    fx = _expected(dname)
    over = json.loads(str(fx["config_json"]))
    o2 = json.loads(json.dumps(over)); o2["encoder"]["from_pt"] = d; o2.setdefault("decoder", {})["from_pt"] = d
    torch.manual_seed(123)
    m = iTransformer(o2, method_name="mlm", log_input=True, loss="poisson_nll", compute_dtype="fp32")
    summary = json.loads(str(fx["state_json"]))
    _check_state(m.state_dict(), summary, skip=("masker",))
    assert any(k.startswith("encoder.region_embeddings.") for k in summary)
    assert any(k.startswith("encoder.depth_embeddings." if dname.endswith("full") else "encoder.embed.transformer.layers.1.") for k in summary)
    torch.manual_seed(5)
    m2 = iTransformer(over, method_name="mlm", log_input=True, loss="poisson_nll", compute_dtype="fp32")
    m2.load_checkpoint(d)
    _check_state(m2.state_dict(), summary, skip=("masker",))


def test_patchtst_reads_the_reference_checkpoint():
    from llm_bci_amd.patchtst import PatchTSTForSpikingActivity
    d = os.path.join(GOLD, "ckpt_patchtst")
    fx = _expected("ckpt_patchtst")
    over = json.loads(str(fx["config_json"]))
    torch.manual_seed(123)
    m = PatchTSTForSpikingActivity(over, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    m.load_checkpoint(d)
    _check_state(m.state_dict(), json.loads(str(fx["state_json"])))
    # the reference pickles its config dicts under a .yaml name (patchtst.py:259,261): from_pt here accepts exactly that
    o2 = json.loads(json.dumps(over)); o2["encoder"]["from_pt"] = d; o2.setdefault("decoder", {})["from_pt"] = d
    m2 = PatchTSTForSpikingActivity(o2, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    _check_state(m2.state_dict(), json.loads(str(fx["state_json"])))


def test_bci_reads_the_reference_checkpoint_including_the_llm():
    from transformers import AutoModelForCausalLM, LlamaConfig
    from llm_bci_amd.bci import BCI
    d = os.path.join(GOLD, "ckpt_bci")
    fx = _expected("ckpt_bci")
    cfg = json.loads(str(fx["config_json"]))
    c2 = json.loads(json.dumps(cfg)); c2["from_pt"] = d
    m = BCI(c2, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")       # bci.py:46,57,78-80,99-104
    _check_state(m.ndt1.state_dict(), json.loads(str(fx["ndt1_state_json"])))
    _check_state(m.projector.state_dict(), json.loads(str(fx["projector_state_json"])))
    _check_state(m.llm.state_dict(), json.loads(str(fx["llm_state_json"])))
    torch.manual_seed(9)
    llm = AutoModelForCausalLM.from_config(LlamaConfig(**json.loads(str(fx["llm_config_json"]))))
    m2 = BCI(cfg, llm=llm, method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True, compute_dtype="fp32")
    m2.load_checkpoint(d)                                                                                      # bci.py:259-264
    _check_state(m2.ndt1.state_dict(), json.loads(str(fx["ndt1_state_json"])))
    _check_state(m2.projector.state_dict(), json.loads(str(fx["projector_state_json"])))
    _check_state(m2.llm.state_dict(), json.loads(str(fx["llm_state_json"])))


def test_the_reverse_direction_was_asserted_against_the_reference():
    r = json.load(open(os.path.join(GOLD, "ckpt_reverse_check.json")))["native_checkpoint_loaded_by_reference"]
    assert r["NDT1"]["reference_from_pt_equal_tensors"] >= 41 and r["NDT1"]["reference_load_checkpoint"] == "equal"
    assert r["iTransformer"]["reference_from_pt_equal_tensors"] > 0 and r["PatchTST"]["reference_load_checkpoint_equal_tensors"] > 0
    assert r["BCI"]["reference_from_pt_equal_tensors"] > 0
    assert r["iTransformer_full"]["reference_from_pt_equal_tensors"] == 49 and r["iTransformer_uni"]["reference_from_pt_equal_tensors"] == 73


@pytest.mark.gpu
def test_ndt1_loaded_from_the_reference_checkpoint_predicts_what_the_reference_predicted():
    d = os.path.join(GOLD, "ckpt_ndt1")
    m, fx = _ndt1()
    m = m.to("cuda")
    m.load_checkpoint(d)
    m.eval()
    b = {k[3:]: torch.from_numpy(fx[k]).to("cuda") for k in fx.files if k.startswith("in_")}
    with torch.no_grad():
        out = m(**b)
    torch.cuda.synchronize()
    assert np.abs(out.preds.cpu().numpy() - fx["preds"]).max() <= 1e-3
    np.testing.assert_allclose(float(out.loss), float(fx["loss"]), rtol=2e-4)


@pytest.mark.gpu
def test_patchtst_loaded_from_the_reference_checkpoint_predicts_what_the_reference_predicted():
    from llm_bci_amd.patchtst import PatchTSTForSpikingActivity
    d = os.path.join(GOLD, "ckpt_patchtst")
    fx = _expected("ckpt_patchtst")
    torch.manual_seed(123)
    m = PatchTSTForSpikingActivity(json.loads(str(fx["config_json"])), method_name="ctc", vocab_size=11, blank_id=0, zero_infinity=True,
                                   compute_dtype="fp32").to("cuda")
    m.load_checkpoint(d)
    m.eval()
    b = {k[3:]: torch.from_numpy(fx[k]).to("cuda") for k in fx.files if k.startswith("in_")}
    with torch.no_grad():
        out = m(**b)
    torch.cuda.synchronize()
    assert np.abs(out.preds.cpu().numpy() - fx["preds"]).max() <= 1e-3
    np.testing.assert_allclose(float(out.loss), float(fx["loss"]), rtol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dname", ["ckpt_itransformer_full", "ckpt_itransformer_uni"])
def test_itransformer_loaded_from_the_reference_checkpoint_predicts_what_the_reference_predicted(dname):
    from llm_bci_amd.itransformer import iTransformer
    d = os.path.join(GOLD, dname)
    fx = _expected(dname)
    torch.manual_seed(123)
    m = iTransformer(json.loads(str(fx["config_json"])), method_name="mlm", log_input=True, loss="poisson_nll", compute_dtype="fp32").to("cuda")
    m.load_checkpoint(d)
    m.eval()
    m.mask_override = torch.from_numpy(fx["raw_mask"])
    b = {k[3:]: (fx[k] if k == "in_neuron_regions" else torch.from_numpy(fx[k]).to("cuda")) for k in fx.files if k.startswith("in_")}
    with torch.no_grad():
        out = m(**b)
    torch.cuda.synchronize()
    assert np.abs(out.preds.cpu().numpy() - fx["preds"]).max() <= 1e-3
    np.testing.assert_allclose(float(out.loss), float(fx["loss"]), rtol=2e-4)
    assert int(out.n_examples) == int(fx["n_examples"])
