"""Bit-exact symbol indexing (north star; SURVEY.md section 4 known answers, 8f N2).  CPU only."""
import unicodedata

import numpy as np

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
    # characters outside the table are dropped, as the reference's _should_keep_symbol does
    assert T.text_to_sequence("가A1나").tolist() == T.text_to_sequence("가나").tolist()
    assert T.text_to_sequence("").tolist() == [1]
    # ids feed the model directly
    assert int(seq.max()) < 80 and int(seq.min()) >= 1
