"""Text -> symbol ids, bit-exact with the reference's indexing (reference text/__init__.py:13-88,
text/korean.py:12-22,149-160, text/symbols.py:14-28).

Host-side integer work (no kernel).  The third-party ``jamo`` package the reference imports is absent here;
its ``hangul_to_jamo`` is restated from the Unicode Hangul-syllable arithmetic (lead 0x1100 + s//588,
vowel 0x1161 + (s%588)//28, tail 0x11A7 + s%28).  Pinned by the reference's own known answers
(text/__init__.py:41, text/cleaners.py:29, text/symbols.py:19-28) and cross-checked against
``unicodedata.normalize('NFD', ...)``.

``tokenize`` runs ``normalize()`` first, as the reference does (text/korean.py:151): numbers, units, English words and
abbreviations, quotes -> Korean words (``text/korean.py`` of this package, pinned by the reference's own outputs in
tests/golden/text_normalize.json).  Symbols that are not in the table after that are dropped exactly as the reference's
``_should_keep_symbol`` drops them.
"""
import re

import numpy as np

from .korean import normalize

PAD = "_"
EOS = "~"
PUNC = "!'(),-.:;?"
SPACE = " "
JAMO_LEADS = "".join(chr(c) for c in range(0x1100, 0x1113))
JAMO_VOWELS = "".join(chr(c) for c in range(0x1161, 0x1176))
JAMO_TAILS = "".join(chr(c) for c in range(0x11A8, 0x11C3))
VALID_CHARS = JAMO_LEADS + JAMO_VOWELS + JAMO_TAILS + PUNC + SPACE
symbols = PAD + EOS + VALID_CHARS                       # 80 symbols (reference text/korean.py:21-22)
_symbol_to_id = {s: i for i, s in enumerate(symbols)}
_id_to_symbol = {i: s for i, s in enumerate(symbols)}
_curly_re = re.compile(r"(.*?)\{(.+?)\}(.*)")

_S_BASE, _S_END = 0xAC00, 0xD7A3


def hangul_to_jamo(text):
    """Precomposed Hangul syllables -> conjoining jamo (lead, vowel[, tail]); other characters pass through."""
    out = []
    for ch in text:
        o = ord(ch)
        if _S_BASE <= o <= _S_END:
            s = o - _S_BASE
            out.append(chr(0x1100 + s // 588))
            out.append(chr(0x1161 + (s % 588) // 28))
            if s % 28:
                out.append(chr(0x11A7 + s % 28))
        else:
            out.append(ch)
    return out


def jamo_to_hangul(tokens):
    """Inverse of ``hangul_to_jamo`` for well-formed lead+vowel(+tail) runs (reference jamo_to_korean)."""
    out, i, n = [], 0, len(tokens)
    while i < n:
        c = tokens[i]
        if c in JAMO_LEADS and i + 1 < n and tokens[i + 1] in JAMO_VOWELS:
            lead, vowel, tail = ord(c) - 0x1100, ord(tokens[i + 1]) - 0x1161, 0
            i += 2
            if i < n and tokens[i] in JAMO_TAILS:
                tail = ord(tokens[i]) - 0x11A7
                i += 1
            out.append(chr(_S_BASE + lead * 588 + vowel * 28 + tail))
        else:
            out.append(c)
            i += 1
    return "".join(out)


def tokenize(text, as_id=False):
    """Reference text/korean.py:149-160: normalize, then jamo tokens + EOS."""
    tokens = hangul_to_jamo(normalize(text))
    if as_id:
        return [_symbol_to_id[t] for t in tokens] + [_symbol_to_id[EOS]]
    return tokens + [EOS]


def _should_keep_symbol(s):
    return s in _symbol_to_id and s != "_" and s != "~"


def _symbols_to_sequence(syms):
    return [_symbol_to_id[s] for s in syms if _should_keep_symbol(s)]


def text_to_sequence(text, as_token=False):
    """Reference text/__init__.py:16-45: ids of the kept symbols, then EOS; int32 array."""
    sequence = []
    while len(text):
        m = _curly_re.match(text)
        if not m:
            sequence += _symbols_to_sequence(tokenize(text))
            break
        sequence += _symbols_to_sequence(tokenize(m.group(1)))
        sequence += _symbols_to_sequence(["@" + s for s in m.group(2).split()])   # ARPAbet: none are in the table
        text = m.group(3)
    sequence.append(_symbol_to_id[EOS])
    if as_token:
        return sequence_to_text(sequence, combine_jamo=True)
    return np.array(sequence, dtype=np.int32)


def sequence_to_text(sequence, skip_eos_and_pad=False, combine_jamo=False):
    """Reference text/__init__.py:48-68."""
    result = []
    for i in sequence:
        i = int(i)
        if i in _id_to_symbol:
            s = _id_to_symbol[i]
            if not skip_eos_and_pad or s not in (EOS, PAD):
                result.append(s)
    return jamo_to_hangul(result) if combine_jamo else "".join(result)
