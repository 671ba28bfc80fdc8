"""Bit-exact symbol indexing (north star; SURVEY.md section 4 known answers, 8f N2).  CPU only."""
import json
import os
import unicodedata

import numpy as np
import pytest

from text2speech_amd import text as T


def test_symbol_table_matches_reference_comment():
    # reference text/symbols.py:19-28 lists every (symbol, id) pair; spot the documented anchors
    assert len(T.symbols) == 80
    want = {"_": 0, "~": 1, "ᄀ": 2, "ᄒ": 20, "ᅡ": 21, "ᅵ": 41, "ᆨ": 42, "ᆫ": 45, "ᆼ": 62, "ᇂ": 68, "!": 69, "'": 70,
            "(": 71, ")": 72, ",": 73, "-": 74, ".": 75, ":": 76, ";": 77, "?": 78, " ": 79}
    for s, i in want.items():
        assert T._symbol_to_id[s] == i, s
    assert T.symbols[2:21] == "".join(chr(c) for c in range(0x1100, 0x1113))


def test_known_answer_from_reference():
    # reference text/cleaners.py:29 and text/__init__.py:41
    assert T.tokenize("존경하는") == ["ᄌ", "ᅩ", "ᆫ", "ᄀ", "ᅧ", "ᆼ", "ᄒ", "ᅡ", "ᄂ", "ᅳ", "ᆫ", "~"]
    seq = T.text_to_sequence("존경하는")
    assert seq.dtype == np.int32
    assert seq.tolist() == [14, 29, 45, 2, 27, 62, 20, 21, 4, 39, 45, 1]


def test_every_syllable_against_unicode_nfd():
    """All 11172 precomposed syllables: our arithmetic == Unicode canonical decomposition."""
    for o in range(0xAC00, 0xD7A4):
        ch = chr(o)
        assert "".join(T.hangul_to_jamo(ch)) == unicodedata.normalize("NFD", ch)
        assert T.jamo_to_hangul(T.hangul_to_jamo(ch)) == ch


def test_punctuation_unknowns_and_roundtrip():
    s = "안녕하세요, 반갑습니다!"
    seq = T.text_to_sequence(s)
    assert seq[-1] == 1 and 73 in seq and 69 in seq and 79 in seq
    assert T.sequence_to_text(seq, skip_eos_and_pad=True, combine_jamo=True) == s
    # Latin letters and digits are spelled out by normalize() (reference text/korean.py:151,163-177) ...
    assert T.text_to_sequence("가A1나").tolist() == T.text_to_sequence("가에이일나").tolist()
    # ... and characters that are still outside the table afterwards are dropped, as the reference's _should_keep_symbol does
    assert T.text_to_sequence("가#나@").tolist() == T.text_to_sequence("가나").tolist()
    assert T.text_to_sequence("").tolist() == [1]
    # ids feed the model directly
    assert int(seq.max()) < 80 and int(seq.min()) >= 1


def _golden():
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_normalize.json")
    return json.load(open(p, encoding="utf-8"))


def test_normalize_bit_exact_vs_reference_outputs():
    """normalize() against the reference's own outputs (tools/gen_golden_text.py imports text/korean.py of the reference):
    the reference's demo inputs (korean.py:335-341) plus digits / separators / signs / decimals, count words, units, English
    words and abbreviations, the phrase table, parenthesised day notes and Hanja."""
    from text2speech_amd.text.korean import normalize
    g = _golden()
    assert len(g["cases"]) >= 100
    for c in g["cases"]:
        assert normalize(c["text"]) == c["normalized"], c["text"]


def test_ids_follow_the_normalized_text_bit_exact():
    """text_to_sequence(raw) == ids of the jamo decomposition of the REFERENCE's normalized string (decomposition pinned by
    test_every_syllable_against_unicode_nfd and the known answers above)."""
    g = _golden()
    n_changed = 0
    for c in g["cases"]:
        want = [T._symbol_to_id[t] for t in T.hangul_to_jamo(c["normalized"]) if T._should_keep_symbol(t)] + [1]
        got = T.text_to_sequence(c["text"])
        assert got.dtype == np.int32 and got.tolist() == want, c["text"]
        n_changed += c["text"].strip() != c["normalized"]
    assert n_changed > 80          # the fixture is about inputs that normalisation rewrites
    # spot values worked by hand from the reference's demo (korean.py:337-340)
    assert T.text_to_sequence("60.3%").tolist() == T.text_to_sequence("육십쩜 삼퍼센트").tolist()
    # a count word selects native numerals digit by digit (the reference's rule, korean.py:279-283): 3,600마리 -> 세천여섯백마리
    assert T.text_to_sequence("오늘(13일) 3,600마리 강아지가").tolist() == T.text_to_sequence("오늘 세천여섯백마리 강아지가").tolist()


def test_normalize_reference_quirks_are_kept():
    from text2speech_amd.text.korean import normalize
    assert normalize("0개") == "영"                      # the count word after a zero is lost (korean.py:264-265)
    assert normalize("12.5개") == "열두개"                # a decimal part in front of a count word is dropped (group 1 only)
    with pytest.raises(ValueError):
        normalize("+5")                                  # int('+') in the reference's digit loop
    # inputs the reference itself rejects raise the same exception type here (recorded by tools/gen_golden_text.py)
    raised = 0
    for u in _golden()["unpinned"]:
        kind = u["why"].split(":")[0]
        if kind in ("ValueError", "SyntaxError", "IndexError"):
            with pytest.raises({"ValueError": ValueError, "SyntaxError": SyntaxError, "IndexError": IndexError}[kind]):
                normalize(u["text"])
            raised += 1
    assert raised >= 3
    # unpinned corner (nltk punkt absent upstream here): single-sentence quotes are re-quoted with ASCII apostrophes
    assert normalize('"저돌"(猪突) 입니다.') == "'저돌' 입니다."
