"""Data front end: tokenizer with private-use-area speech units, SFT and CPT sample construction, loaders.

No tokenizer file, dataset or fixture with token ids ships with the reference, and tiktoken / torchtune / sardalign are not on the
image, so token-level results are **parity unpinned**; what is pinned: the reference's own property tests of the per-sample
generators (``/root/reference/tests/test_cpt_deterministic_rng.py``, mirrored below case by case), hand-derived encodings on
a toy merge table, and the rules the reference states in its sources (masking, deduplication, truncation, padding)."""
import json
import os

import numpy as np
import pytest
import torch

from ssi.constants import SEED
from ssi.data import (SFTDataset, TextCompletionDataset, concatenate_speech_text, get_span_idxs_binomial, interleave, padded_collate_sft,
                      setup_sft_data, setup_text_completion_data)
from ssi.tokenizer import (CL100K_PATTERN_PUA, LLAMA3_SPECIAL_TOKENS, MODALITY_TOKEN_SPEECH, MODALITY_TOKEN_TEXT, Llama3TokenizerPUA, Message,
                           deduplicate_units, dsu2pua, dump_tiktoken_bpe, pua2dsu, setup_llama3_tokenizer, units_to_text, validate_messages)

N_UNITS = 50
MERGES = [b"th", b"he", b"the", b" t", b" the", b"in", b"\n\n", b"er", b"us", b"user", b"as", b"sy", b"st", b"ss", b" a", b"an", b"and", b" and"]


def toy_ranks(n_units=N_UNITS, modality=True):
    """256 bytes, a few merges, then — as extend_tiktoken appends them — one token per speech unit and the two modality tokens."""
    ranks = {bytes([i]): i for i in range(256)}
    for m in MERGES:
        ranks[m] = len(ranks)
    for k in range(n_units):
        ranks[dsu2pua(k).encode()] = len(ranks)
    if modality:
        ranks[MODALITY_TOKEN_TEXT.encode()] = len(ranks)
        ranks[MODALITY_TOKEN_SPEECH.encode()] = len(ranks)
    return ranks


N_TXT = 256 + len(MERGES)
UNIT0 = N_TXT
ID_TEXT, ID_SPEECH = N_TXT + N_UNITS, N_TXT + N_UNITS + 1
BASE = N_TXT + N_UNITS + 2


@pytest.fixture()
def tok(tmp_path):
    path = tmp_path / "tokenizer.model"
    dump_tiktoken_bpe(toy_ranks(), path)
    t, special = setup_llama3_tokenizer(path, max_seq_len=None, verbose=False)
    return t


# ---- reference tests/test_cpt_deterministic_rng.py, case by case -----------------------------------------------------------
def make_rng(seed, epoch, index):
    return np.random.default_rng((seed, epoch, index))


@pytest.mark.parametrize("seed, epoch, index", [(42, 0, 7), (42, 0, 0), (0, 0, 0), (99999, 10, 500), (42, 3, 12345)])
def test_span_idxs_reproducible(seed, epoch, index):
    a = get_span_idxs_binomial(5, 0.4, 50, rng=make_rng(seed, epoch, index))
    assert a == get_span_idxs_binomial(5, 0.4, 50, rng=make_rng(seed, epoch, index))


def test_span_idxs_differ_by_index_and_epoch_but_not_by_order():
    n, p, L = 5, 0.4, 50
    assert get_span_idxs_binomial(n, p, L, rng=make_rng(42, 0, 0)) != get_span_idxs_binomial(n, p, L, rng=make_rng(42, 0, 1))
    assert get_span_idxs_binomial(n, p, L, rng=make_rng(42, 0, 7)) != get_span_idxs_binomial(n, p, L, rng=make_rng(42, 1, 7))
    seq = {i: get_span_idxs_binomial(n, p, L, rng=make_rng(42, 3, i)) for i in range(20)}
    order = [13, 7, 2, 18, 0, 15, 9, 4, 11, 19, 6, 1, 16, 3, 14, 8, 17, 5, 12, 10]
    shuffled = {i: get_span_idxs_binomial(n, p, L, rng=make_rng(42, 3, i)) for i in order}
    assert seq == shuffled


def test_span_idxs_boundary_invariants():
    for idx in range(50):
        spans = get_span_idxs_binomial(5, 0.4, 100, rng=make_rng(42, 0, idx))
        assert spans[0] == 0 and spans[-1] == 100 and spans == sorted(spans) and len(spans) >= 2
    for L in (1, 2, 5, 10, 100, 1000):
        spans = get_span_idxs_binomial(5, 0.4, L, rng=make_rng(42, 0, 0))
        assert spans[0] == 0 and spans[-1] == L


# ---- units and tokenizer -------------------------------------------------------------------------------------------------------
def test_units_are_private_use_characters():
    assert dsu2pua(0) == "" and dsu2pua(4999) == chr(0xE000 + 4999) and pua2dsu(dsu2pua(6399)) == 6399
    assert ord(dsu2pua(6400)) == 0xF0000 and pua2dsu(dsu2pua(8191)) == 8191       # 8192-unit codebooks leave the BMP area
    with pytest.raises(ValueError):
        dsu2pua(-1)
    assert deduplicate_units([5, 5, 6, 6, 6, 5, 7, 7]) == [5, 6, 5, 7] == deduplicate_units(np.array([5, 5, 6, 6, 6, 5, 7, 7]))
    assert deduplicate_units([]) == [] == deduplicate_units(np.array([], dtype=np.int64))


def test_pattern_is_the_cl100k_pattern_with_private_use_characters_singled_out():
    import regex
    assert CL100K_PATTERN_PUA.endswith(r"|\p{Co}") and CL100K_PATTERN_PUA.count(r"\p{Co}") == 3
    pat = regex.compile(CL100K_PATTERN_PUA)
    s = "Hello world's" + dsu2pua(3) + dsu2pua(4) + " 12345 ok"
    assert pat.findall(s) == ["Hello", " world", "'s", dsu2pua(3), dsu2pua(4), " ", "123", "45", " ok"]
    # a space run before a unit: the lookahead leaves the last blank to stand alone, as it does before any other non-blank
    assert pat.findall("a  " + dsu2pua(1)) == ["a", " ", " ", dsu2pua(1)]


def test_setup_numbers_the_specials_after_the_ranked_tokens(tok):
    assert len(LLAMA3_SPECIAL_TOKENS) == 256 and tok.base_vocab_size == BASE and tok.vocab_size == BASE + 256
    assert (tok.bos_id, tok.eos_id, tok.pad_id, tok.start_header_id, tok.end_header_id, tok.eom_id, tok.eot_id, tok.python_tag) == \
        tuple(BASE + i for i in (0, 1, 4, 6, 7, 8, 9, 10))
    assert tok.special_tokens["<|reserved_special_token_2|>"] == BASE + 13 and tok.special_tokens["<|image|>"] == BASE + 11
    assert tok.n_units == N_UNITS and tok.unit_ids.tolist() == list(range(UNIT0, UNIT0 + N_UNITS))


def test_setup_without_a_file_falls_back_to_the_layout_and_refuses_text(tmp_path):
    from ssi.llama_configs import configllama3_2_1b
    import copy
    lc = copy.deepcopy(configllama3_2_1b)
    lc.n_dsus, lc.modality_tokens = 5000, True
    t, special = setup_llama3_tokenizer(tmp_path / "missing.model", max_seq_len=2048, llama_config=lc)
    assert t.pad_id == 128000 + 5000 + 2 + 4 and special["<|eot_id|>"] == t.pad_id + 5
    with pytest.raises(RuntimeError, match="tokenizer.path"):
        t.encode("text")
    with pytest.raises(ValueError):
        setup_llama3_tokenizer(None)


def test_encode_byte_pair_merges_units_and_modality_tokens(tok):
    # lowest-ranked adjacent pair first: "there" -> th|e|r|e -> the|r|e -> the|re? ("re" is not in the table) -> the, er? no: after "the" the
    # remaining pairs are (the,r) and (r,e), neither ranked, so "er" never forms: [the, r, e]
    assert tok.encode("there", add_bos=False, add_eos=False) == [256 + 2, ord("r"), ord("e")]
    assert tok.encode(" the user", add_bos=False, add_eos=False) == [256 + 4, 32, 256 + 9]
    assert tok.encode("", add_bos=True, add_eos=True) == [tok.bos_id, tok.eos_id]
    units = [1, 2, 2, 49]
    text = MODALITY_TOKEN_SPEECH + units_to_text(units) + MODALITY_TOKEN_TEXT + " the"
    ids = tok.encode(text, add_bos=False, add_eos=False)
    assert ids == [ID_SPEECH, UNIT0 + 1, UNIT0 + 2, UNIT0 + 2, UNIT0 + 49, ID_TEXT, 256 + 4]
    assert tok.encode_units(units) == ids[1:5]
    assert tok.decode(ids) == text
    # special-token strings inside text are ordinary text (allowed_special = {}), never the special id
    assert tok.eot_id not in tok.encode("<|eot_id|>", add_bos=False, add_eos=False)
    # a unit the table does not list falls apart into its UTF-8 bytes, like any unknown character
    assert tok.encode(dsu2pua(N_UNITS), add_bos=False, add_eos=False) == list(dsu2pua(N_UNITS).encode())
    assert tok.decode([tok.bos_id, 256 + 2, tok.eos_id, 65]) == "the" and tok.decode([tok.bos_id, 65], skip_special_tokens=False) == "<|begin_of_text|>A"


def test_long_runs_are_cut_before_pre_tokenisation():
    from ssi.tokenizer.llama3_pua import _split_long_repetitions
    assert _split_long_repetitions("aaaa  bbbbbbb", 3) == ["aaa", "a  bbb", "bbb", "b"]
    assert _split_long_repetitions("ab", 3) == ["ab"]


def test_messages_headers_masks_eos_and_truncation(tok):
    msgs = [Message("system", " be brief ", masked=True), Message("user", "the"), Message("assistant", "he")]
    validate_messages(msgs)
    tokens, mask = tok.tokenize_messages(msgs)
    H = lambda role: [tok.start_header_id] + tok.encode(role, False, False) + [tok.end_header_id, 256 + 6]   # "\n\n" is one merge
    sys_body = tok.encode("be brief", False, False)       # content is stripped
    expect = [tok.bos_id] + H("system") + sys_body + [tok.eot_id] + H("user") + [256 + 2, tok.eot_id] + H("assistant") + [256 + 1, tok.eot_id, tok.eos_id]
    assert tokens == expect
    n_sys = len(H("system")) + len(sys_body) + 1
    assert mask == [True] + [True] * n_sys + [False] * (len(expect) - n_sys - 2) + [True]      # BOS, system prompt and EOS carry no loss
    t2, m2 = tok.tokenize_messages(msgs, add_eos=False)
    assert t2 == expect[:-1] and m2 == mask[:-1]
    # truncation: stop adding messages once max_seq_len is reached, cut, and end on EOS (only when one was asked for)
    tok.max_seq_len = 12
    t3, m3 = tok.tokenize_messages(msgs)
    assert len(t3) == 12 and t3[:11] == expect[:11] and t3[-1] == tok.eos_id and m3[-1] is True
    t4, _ = tok.tokenize_messages(msgs, add_eos=False)
    assert t4 == expect[:12]
    # eom instead of eot, ipython tag
    tok.max_seq_len = None
    t5, _ = tok.tokenize_messages([Message("user", "a"), Message("assistant", "b", eot=False), Message("ipython", "c", ipython=True)], add_eos=False)
    assert tok.eom_id in t5 and tok.python_tag in t5
    for bad in ([Message("user", "a")], [Message("user", "a"), Message("user", "b")], [Message("assistant", "a"), Message("user", "b")],
                [Message("user", "a"), Message("system", "b")]):
        with pytest.raises(ValueError):
            validate_messages(bad)


# ---- SFT ---------------------------------------------------------------------------------------------------------------------------
ROWS = [{"ID": f"utt{i}", "speech_tokens": [(i + j // 3) % N_UNITS for j in range(20 + 7 * i)], "transcript": "the user and the hen " * (1 + i % 3)}
        for i in range(12)]
SYS = "You will act as an automatic speech recognition (ASR) system. "


def sft(tok, **kw):
    args = dict(source=ROWS, model_tokenizer=tok, deduplicate=True, use_modality_tokens=True, train_on_input=True,
                column_map={"input": "speech_tokens", "output": "transcript"}, new_system_prompt=SYS)
    args.update(kw)
    return SFTDataset(**args)


def test_sft_sample_layout_labels_and_switches(tok):
    ds = sft(tok, additional_keys=["ID"])
    assert len(ds) == 12
    s = ds[3]
    assert s["ID"] == "utt3" and len(s["tokens"]) == len(s["labels"]) == len(s["mask"])
    units = deduplicate_units(ROWS[3]["speech_tokens"])
    span = [ID_SPEECH] + [UNIT0 + u for u in units] + [ID_TEXT]
    toks = s["tokens"]
    at = next(i for i in range(len(toks)) if toks[i:i + len(span)] == span)          # the user message body
    assert toks[at - 1] == 256 + 6 and toks[at + len(span)] == tok.eot_id and toks[0] == tok.bos_id and toks[-1] == tok.eos_id
    labels = np.array(s["labels"]); tokens = np.array(toks); mask = np.array(s["mask"])
    assert (labels[mask] == -100).all() and (labels[~mask] == tokens[~mask]).all()
    assert mask[0] and mask[-1] and not mask[at:at + len(span)].any()                 # train_on_input: the speech span carries loss
    n_sys = len(tok.tokenize_message(Message("system", SYS)))
    assert mask[1:1 + n_sys].all() and not mask[1 + n_sys:-1].any()                   # exactly BOS + system prompt + EOS are ignored
    # switches
    s_nodedup = sft(tok, deduplicate=False)[3]["tokens"]
    assert sum(UNIT0 <= t < UNIT0 + N_UNITS for t in s_nodedup) == len(ROWS[3]["speech_tokens"])
    s_nomod = sft(tok, use_modality_tokens=False)[3]["tokens"]
    assert ID_SPEECH not in s_nomod and ID_TEXT not in s_nomod
    s_mask_in = sft(tok, train_on_input=False)[3]
    assert all(l == -100 for l in s_mask_in["labels"][at:at + len(span)])
    s_inf = sft(tok, inference=True)[3]["tokens"]                                    # generation: empty assistant turn, no EOS
    assert s_inf[-1] == tok.eot_id and s_inf[-2] == 256 + 6 and tok.eos_id not in s_inf
    with pytest.raises(TypeError):
        ds.deduplicate = "yes"
    with pytest.raises(ValueError, match="reserved keys"):
        sft(tok, source=[dict(ROWS[0], tokens=[1])])
    assert len(sft(tok, filter_fn=lambda r: r["ID"] != "utt0")) == 11 and len(sft(tok, n_samples=5)) == 5


def _cfg(d):
    from ssi.config import OmegaConf
    return OmegaConf.create(d)


def test_setup_sft_data_from_a_local_json_file_pads_and_packs(tok, tmp_path, monkeypatch):
    monkeypatch.setenv("HF_DATASETS_OFFLINE", "1")
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    data_file = tmp_path / "train.jsonl"
    data_file.write_text("\n".join(json.dumps(r) for r in ROWS))
    tok.max_seq_len = 128
    node = {"dataset": {"source": "json", "data_files": str(data_file), "split": "train", "inference": False, "deduplicate": True, "filter_fn": None,
                        "train_on_input": True, "column_map": {"input": "speech_tokens", "output": "transcript"}, "new_system_prompt": SYS,
                        "image_dir": None, "use_modality_tokens": True, "n_samples": None, "fixed_len": True, "additional_keys": ["ID"]},
            "dataloader": {"batch_size": 4, "drop_last": True, "num_workers": 0}, "shuffle": False, "packed": False}
    loader, sampler = setup_sft_data(_cfg(node), tok)
    assert len(loader) == 3 and sampler.seed == SEED
    ref = sft(tok)
    for b, batch in enumerate(loader):
        assert batch["tokens"].dtype == batch["labels"].dtype == torch.int64 and batch["tokens"].shape == batch["labels"].shape
        assert batch["ID"] == [f"utt{4 * b + i}" for i in range(4)]
        for i in range(4):
            s = ref[4 * b + i]
            n = len(s["tokens"])
            assert batch["tokens"][i, :n].tolist() == s["tokens"] and batch["labels"][i, :n].tolist() == s["labels"]
            assert (batch["tokens"][i, n:] == tok.pad_id).all() and (batch["labels"][i, n:] == -100).all()
        assert batch["tokens"].shape[1] == max(len(ref[4 * b + i]["tokens"]) for i in range(4))
    # packed: rows of max_seq_len tokens with per-document positions (the reference raises NotImplementedError here)
    node["packed"] = True
    tok.max_seq_len = 256
    node["dataset"]["additional_keys"] = []
    node["dataloader"]["batch_size"] = 2
    loader, _ = setup_sft_data(_cfg(node), tok)
    batch = next(iter(loader))
    assert batch["tokens"].shape == (2, 256) and batch["input_pos"].shape == (2, 256)
    first, second = ref[0]["tokens"], ref[1]["tokens"]
    assert len(first) + len(second) <= 256
    assert batch["tokens"][0, :len(first) + len(second)].tolist() == first + second
    assert batch["input_pos"][0, :len(first) + 2].tolist() == list(range(len(first))) + [0, 1]     # positions restart with each sample
    assert batch["seq_lens"][0][:2].tolist() == [len(first), len(second)]


# ---- CPT -------------------------------------------------------------------------------------------------------------------------
def cpt_rows(n=8, words=30):
    rows = []
    for i in range(n):
        w = [("the", "user", "and", "hen", "ant")[(i + j) % 5] for j in range(words)]
        starts = [0.4 * j for j in range(words)]
        ends = [0.4 * j + 0.3 for j in range(words)]
        units = [(i + j // 4) % N_UNITS for j in range(int(0.4 * words * 50) + 10)]   # 50 units per second
        rows.append({"tokenized": w, "aligned_start_times": starts, "aligned_end_times": ends, "speech_tokens": units})
    return rows


IKW = {"sampling_rate": 16000, "downsampling_ratio": 320, "mean_seq_len_tokens": 5.0, "binom_prob": 0.4}


def test_concatenated_sequences(tok):
    rows = cpt_rows()
    for seq_type, text_first in (("concatenated_txt_dsu", True), ("concatenated_dsu_txt", False)):
        ds = TextCompletionDataset(tok, rows, sequence_type=seq_type, deduplicate=True, use_modality_tokens=True)
        s = ds[2]
        assert s["tokens"] == s["labels"] and s["tokens"][0] == tok.bos_id and s["tokens"][-1] == tok.eos_id
        body = s["tokens"][1:-1]
        units = [UNIT0 + u for u in deduplicate_units(rows[2]["speech_tokens"])]
        text = tok.encode(" " + " ".join(rows[2]["tokenized"]), False, False)
        # "<text> words <speech> units"; every part is joined by one blank, which stands alone in front of a unit or modality token
        expect = ([ID_TEXT] + text + [32, ID_SPEECH, 32] + units) if text_first else ([ID_SPEECH, 32] + units + [32, ID_TEXT] + text)
        assert body == expect
    plain = concatenate_speech_text(rows[0], deduplicate=False, use_modality_tokens=False, rng=None, start_with_text=True)
    assert plain == " ".join(rows[0]["tokenized"]) + " " + units_to_text(rows[0]["speech_tokens"])
    ds = TextCompletionDataset(tok, rows, sequence_type="concatenated_txt_dsu", deduplicate=True, use_modality_tokens=True, add_eos=False)
    assert ds[0]["tokens"][-1] != tok.eos_id
    with pytest.raises(ValueError):
        TextCompletionDataset(tok, rows, sequence_type="dsu_only", deduplicate=True, use_modality_tokens=True)
    with pytest.raises(ValueError, match="interleave_kwargs"):
        TextCompletionDataset(tok, rows, sequence_type="interleaved", deduplicate=True, use_modality_tokens=True)


def test_interleaved_sequences_cover_the_utterance_once_and_depend_on_seed_epoch_index_only(tok):
    rows = cpt_rows()
    ds = TextCompletionDataset(tok, rows, sequence_type="interleaved", deduplicate=False, use_modality_tokens=True, interleave_kwargs=IKW)
    a = [ds[i]["tokens"] for i in range(len(ds))]
    b = {i: ds[i]["tokens"] for i in (5, 1, 7, 0, 3, 2, 6, 4)}
    assert all(a[i] == b[i] for i in range(len(ds)))                      # order of access does not matter
    ds.set_epoch(1)
    assert any(ds[i]["tokens"] != a[i] for i in range(len(ds)))            # the epoch does
    ds.set_epoch(0)
    assert ds[4]["tokens"] == a[4]
    # the same draw, by hand: first the coin, then the span lengths
    rng = np.random.default_rng((SEED, 0, 4))
    start_with_text = bool(rng.choice([True, False], p=[0.5, 0.5]))
    spans = get_span_idxs_binomial(5, 0.4, 30, rng=rng)
    prompt = interleave(rows[4], False, True, rng=np.random.default_rng((SEED, 0, 4)), **IKW)
    pieces = prompt.split(" ")
    assert pieces[0] == (MODALITY_TOKEN_TEXT if start_with_text else MODALITY_TOKEN_SPEECH)
    words = [p for p in pieces if p in ("the", "user", "and", "hen", "ant")]
    n_text_words = sum(b - a for k, (a, b) in enumerate(zip(spans[:-1], spans[1:])) if (k % 2 == 0) == start_with_text)
    assert len(words) == n_text_words
    # unit spans are the slices the word times select: 50 units per second, int() of the product
    k0 = 0 if not start_with_text else 1
    a0, b0 = spans[k0], spans[k0 + 1]
    lo, hi = int(rows[4]["aligned_start_times"][a0] * 16000 / 320), int(rows[4]["aligned_end_times"][b0 - 1] * 16000 / 320)
    assert units_to_text(rows[4]["speech_tokens"][lo:hi]) in prompt
    assert tok.encode(prompt, add_bos=True, add_eos=True) == a[4]
    # truncation to max_seq_len - 1 without forcing an EOS
    tok.max_seq_len = 40
    short = ds[4]["tokens"]
    assert short == a[4][:39]
    tok.max_seq_len = None


def test_setup_text_completion_data_loader(tok):
    rows = cpt_rows(10)
    node = {"dataset": {"source": rows, "split": None, "sequence_type": "interleaved", "interleave_kwargs": IKW, "deduplicate": True,
                        "use_modality_tokens": True, "add_eos": True, "n_samples": None, "fixed_len": True, "tokenized_key": None,
                        "alignment_start_time_key": None, "alignment_end_time_key": None, "speech_tokens_key": None},
            "dataloader": {"batch_size": 4, "drop_last": False, "num_workers": 0}, "shuffle": True, "packed": False}

    class Node(dict):   # attribute access like a config node, keeping the in-memory rows as they are
        __getattr__ = dict.__getitem__
    cfg = Node(node)
    cfg["dataset"], cfg["dataloader"] = Node(node["dataset"]), Node(node["dataloader"])
    cfg["dataset"]["interleave_kwargs"] = Node(IKW)
    loader, sampler = setup_text_completion_data(cfg, tok)
    sampler.set_epoch(0)
    batches = list(loader)
    assert [b["tokens"].shape[0] for b in batches] == [4, 4, 2]
    assert all((b["labels"][b["tokens"] == tok.pad_id] == -100).all() for b in batches)
    order = list(iter(sampler))
    assert sorted(order) == list(range(10)) and order != list(range(10))       # shuffled with the project seed
    ds = loader.dataset
    n0 = len(ds[order[0]]["tokens"])
    assert batches[0]["tokens"][0, :n0].tolist() == ds[order[0]]["tokens"]


def test_collate_pads_tokens_and_labels_and_passes_extra_keys():
    out = padded_collate_sft([{"tokens": [1, 2, 3], "labels": [4, 5, 6], "ID": "a"}, {"tokens": [7], "labels": [10], "ID": "b"}], padding_idx=0,
                             ignore_idx=-100, additional_keys=["ID"])
    assert out["tokens"].tolist() == [[1, 2, 3], [7, 0, 0]] and out["labels"].tolist() == [[4, 5, 6], [10, -100, -100]] and out["ID"] == ["a", "b"]
