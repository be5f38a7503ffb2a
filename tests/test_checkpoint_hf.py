"""HF-format checkpoint I/O: key map + q/k row permutation + shard/index layout.  CPU only.
The permutation is pinned against the HF cross-check of the oracle (oracle/hf_crosscheck.py: an HF LlamaForCausalLM built from a
local config and fed the converted weights reproduces the torchtune-semantics oracle), the layout against what the reference
writes (model-0000i-of-0000n.safetensors + model.safetensors.index.json, /root/reference/ssi/checkpoint.py:372-406)."""
import json
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from ssi.checkpoint import (FullModelHFCheckpointer, TuneCheckpointer, discover_safetensor_files, hf_to_tune, make_checkpointer,  # noqa: E402
                            tune_to_hf)
from ssi.constants import MODEL_KEY  # noqa: E402
from ssi.llama_configs import ModelCheckpointExpectations  # noqa: E402

PARAMS = dict(vocab_size=515, num_layers=2, num_heads=8, num_kv_heads=2, embed_dim=128, max_seq_len=64, intermediate_dim=256)
CONV = dict(num_heads=8, num_kv_heads=2, dim=128)


def _tune_sd():
    from oracle import hf_crosscheck as hx
    return hx.seeded_state_dict(PARAMS, 5)


def _write_hf_dir(d: Path, sd_hf: dict, shards: int = 2, vocab: int = 515) -> None:
    from safetensors.torch import save_file
    d.mkdir(parents=True, exist_ok=True)
    keys = sorted(sd_hf)
    per = (len(keys) + shards - 1) // shards
    for i in range(shards):
        part = {k: sd_hf[k].contiguous() for k in keys[i * per:(i + 1) * per]}
        save_file(part, str(d / f"model-{i + 1:05}-of-{shards:05}.safetensors"), metadata={"format": "pt"})
    (d / "config.json").write_text(json.dumps({"num_attention_heads": 8, "num_key_value_heads": 2, "hidden_size": 128, "num_hidden_layers": 2,
                                               "vocab_size": vocab, "tie_word_embeddings": True}))
    (d / "tokenizer.json").write_text("{}")


def test_key_map_and_permutation_round_trip_and_match_the_hf_crosscheck():
    from oracle import hf_crosscheck as hx
    sd = _tune_sd()
    hf = tune_to_hf(sd, **CONV)
    assert "model.embed_tokens.weight" in hf and "model.layers.1.self_attn.o_proj.weight" in hf and "model.norm.weight" in hf
    assert "model.layers.0.post_attention_layernorm.weight" in hf and "model.layers.0.mlp.gate_proj.weight" in hf
    back = hf_to_tune(hf, **CONV)
    assert back.keys() == sd.keys() and all(torch.equal(back[k], sd[k]) for k in sd)
    # same permutation as the one the oracle's HF cross-check uses to feed HF-Llama (which agreed with the oracle to 1e-6)
    q = sd["layers.0.attn.q_proj.weight"]
    assert torch.equal(hf["model.layers.0.self_attn.q_proj.weight"], hx.tune_to_hf_qk(q, 8))
    k = sd["layers.1.attn.k_proj.weight"]
    assert torch.equal(hf["model.layers.1.self_attn.k_proj.weight"], hx.tune_to_hf_qk(k, 2))
    assert torch.equal(hf["model.layers.0.mlp.down_proj.weight"], sd["layers.0.mlp.w2.weight"])  # everything else is copied
    # tied head and derived buffers are dropped on the way in; unknown keys are an error
    extra = dict(hf)
    extra["lm_head.weight"] = hf["model.embed_tokens.weight"]
    extra["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.zeros(8)
    assert hf_to_tune(extra, **CONV).keys() == sd.keys()
    with pytest.raises(KeyError):
        hf_to_tune({"model.layers.0.bogus.weight": torch.zeros(1)}, **CONV)


def test_hf_checkpointer_loads_shards_and_writes_the_reference_layout(tmp_path):
    sd = _tune_sd()
    src, out = tmp_path / "hf_model", tmp_path / "out"
    _write_hf_dir(src, tune_to_hf(sd, **CONV), shards=2)
    assert discover_safetensor_files(src) == ["model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors"]
    ck = make_checkpointer(checkpoint_dir=str(src), checkpoint_files=None, output_dir=str(out), training_state_checkpoint=None,
                           model_expectations=ModelCheckpointExpectations("tiny", 2, 2, 128, 515))
    assert isinstance(ck, FullModelHFCheckpointer)
    loaded = ck.load_checkpoint()[MODEL_KEY]
    assert loaded.keys() == sd.keys() and all(torch.equal(loaded[k], sd[k]) for k in sd)
    step_dir = ck.save_model_checkpoint({k: v + 1 for k, v in loaded.items()}, 7)
    assert step_dir == out / "step_7"
    assert sorted(p.name for p in step_dir.iterdir()) == ["config.json", "model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors",
                                                         "model.safetensors.index.json", "tokenizer.json"]
    index = json.loads((step_dir / "model.safetensors.index.json").read_text())
    assert index["metadata"]["total_size"] == sum(v.numel() * v.element_size() for v in sd.values())
    assert set(index["weight_map"]) == set(tune_to_hf(sd, **CONV))
    # the written directory is itself a loadable HF checkpoint and holds the updated weights
    again = FullModelHFCheckpointer(step_dir, None, output_dir=tmp_path / "out2").load_checkpoint()[MODEL_KEY]
    assert all(torch.equal(again[k], sd[k] + 1) for k in sd)
    # training state: schema-v1 file at the output root, merged into the next load
    path = ck.save_training_state(optimizer_state_dict={"state": {}}, lr_scheduler_state_dict=None, global_step=7, seed=1,
                                  training_hparams={"lr": 1.0}, consumed_samples=56, cumulative_metrics={"tokens": 9})
    resumed = FullModelHFCheckpointer(step_dir, None, output_dir=tmp_path / "out3", training_state_checkpoint=path).load_checkpoint()
    assert resumed["global_step"] == 7 and resumed["consumed_samples"] == 56


def test_hf_checkpointer_validation_errors(tmp_path):
    sd_hf = tune_to_hf(_tune_sd(), **CONV)
    src = tmp_path / "hf_model"
    _write_hf_dir(src, sd_hf, shards=1, vocab=999)
    with pytest.raises(ValueError, match="vocab_size"):
        FullModelHFCheckpointer(src, None, output_dir=tmp_path / "o", model_expectations=ModelCheckpointExpectations("tiny", 1, 2, 128, 515))
    with pytest.raises(ValueError, match="shard"):
        FullModelHFCheckpointer(src, None, output_dir=tmp_path / "o", model_expectations=ModelCheckpointExpectations("tiny", 4, 2, 128, 999))
    with pytest.raises(ValueError, match="must not lie inside"):
        FullModelHFCheckpointer(src, None, output_dir=src / "sub")
    with pytest.raises(FileNotFoundError):
        FullModelHFCheckpointer(tmp_path / "nope", None, output_dir=tmp_path / "o")
    (src / "ft-model-00001-of-00001.safetensors").write_bytes((src / "model-00001-of-00001.safetensors").read_bytes())
    with pytest.raises(ValueError, match="Ambiguous"):
        discover_safetensor_files(src)
    with pytest.raises(ValueError, match="Weight map"):
        FullModelHFCheckpointer(src, ["model-00001-of-00001.safetensors"], output_dir=tmp_path / "o2").save_full_model({MODEL_KEY: {}}, tmp_path / "o2" / "x")
    # a directory without config.json is not an HF model directory: the single-file torchtune-key checkpointer is used
    assert type(make_checkpointer(checkpoint_dir=str(tmp_path / "o"), output_dir=str(tmp_path / "o4"))) is TuneCheckpointer


def test_missing_weights_raise_unless_random_init_is_asked_for(tmp_path):
    """A checkpoint_dir with neither config.json nor model.safetensors is an error (reference: ssi/checkpoint.py:263-264), not a
    silent random initialisation; ``allow_random_init`` is the explicit opt-in."""
    ck = make_checkpointer(checkpoint_dir=str(tmp_path / "typo"), output_dir=str(tmp_path / "o"))
    with pytest.raises(FileNotFoundError, match="allow_random_init"):
        ck.load_checkpoint()
    ck = make_checkpointer(checkpoint_dir=str(tmp_path / "typo"), output_dir=str(tmp_path / "o"), allow_random_init=True)
    assert ck.load_checkpoint()[MODEL_KEY] is None
    with pytest.raises(FileNotFoundError, match="Recipe checkpoint"):
        make_checkpointer(checkpoint_dir=str(tmp_path / "typo"), output_dir=str(tmp_path / "o"), allow_random_init=True,
                          training_state_checkpoint=str(tmp_path / "nope.pt"))


def test_training_state_round_trips_through_the_restricted_loader(tmp_path):
    """training_state.pt holds tensors and plain containers only: it loads with weights_only=True, carries the three generator states
    (python, torch_cpu, and NumPy's as tensors under a key of its own) and restores them exactly.  A file whose RNG section holds a
    raw NumPy array (what the reference writes under ``numpy_global``) is refused with a message that says so."""
    import random

    import numpy as np
    from ssi.checkpoint import TuneCheckpointer, restore_rng_states
    random.seed(5), np.random.seed(6), torch.manual_seed(7)
    np.random.standard_normal(3)  # leaves a cached gaussian in the legacy generator
    ck = TuneCheckpointer(checkpoint_dir=None, output_dir=str(tmp_path), allow_random_init=True)
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.ones(3))], lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda n: 1.0)
    path = ck.save_training_state(optimizer_state_dict=opt.state_dict(), lr_scheduler_state_dict=sched.state_dict(), global_step=4, seed=1,
                                  training_hparams={"batch_size": 2}, consumed_samples=8,
                                  cumulative_metrics={"tokens_train_total": 10, "token_type_counts": {"text": 3}, "wall_clock_seconds": 1.5})
    want = (random.random(), np.random.standard_normal(2).tolist(), np.random.randint(0, 100), torch.rand(2))
    state = torch.load(path, map_location="cpu", weights_only=True)
    assert set(state["rng_state"]) >= {"python", "numpy_global_tensors", "torch_cpu"} and state["global_step"] == 4
    random.seed(0), np.random.seed(0), torch.manual_seed(0)
    restore_rng_states(state["rng_state"])
    got = (random.random(), np.random.standard_normal(2).tolist(), np.random.randint(0, 100), torch.rand(2))
    assert got[:3] == want[:3] and torch.equal(got[3], want[3])
    resumed = TuneCheckpointer(checkpoint_dir=None, output_dir=str(tmp_path), allow_random_init=True, training_state_checkpoint=path).load_checkpoint()
    assert resumed["consumed_samples"] == 8 and resumed["cumulative_metrics"]["token_type_counts"] == {"text": 3}
    # the previous key (round-2 files: tensor form under the reference's name) still restores
    legacy = dict(state["rng_state"])
    legacy["numpy_global"] = legacy.pop("numpy_global_tensors")
    random.seed(0), np.random.seed(0), torch.manual_seed(0)
    restore_rng_states(legacy)
    assert (random.random(), np.random.standard_normal(2).tolist(), np.random.randint(0, 100)) == want[:3]
    # a foreign writer's file: ndarray inside -> refused by the restricted loader, with a clear message
    from ssi.checkpoint import load_training_state
    foreign = tmp_path / "foreign_training_state.pt"
    torch.save({"rng_state": {"numpy_global": np.random.get_state()}}, foreign)
    with pytest.raises(RuntimeError, match="restricted loader"):
        load_training_state(str(foreign))
