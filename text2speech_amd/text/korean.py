"""Korean text normalisation: the step the reference runs in front of the jamo decomposition
(reference text/korean.py:149-152 ``tokenize`` -> ``normalize``, :163-326; lookup tables text/ko_dictionary.py).

Host-side string work, no kernel.  Restated from the reference's behaviour and held bit-exact to the reference's own outputs
on 119 inputs (tests/golden/text_normalize.json, written by tools/gen_golden_text.py, which imports the reference):

    strip -> drop "(13일)"-style day notes and parenthesised Hanja -> phrase table -> English word table -> spell out
    all-capital words letter by letter -> re-quote quoted text sentence by sentence -> units -> numbers with a counting
    word (native numerals: 한, 두, 세 ... 열, 스물 ...) -> remaining numbers (Sino-Korean: 일, 이, 삼 ... 십, 백, 천, 만, 억 ...)

The two word tables are data, kept in ``ko_dictionary.json`` exactly as the reference's dict objects hold them (insertion
order matters: the phrase table is applied as ONE alternation, first listed key first).

Unpinned corner (documented): inside quotation marks the reference calls ``nltk.sent_tokenize`` (the punkt
model, an absent and unpinned third-party dependency); here a quoted text is split after ``.``, ``!`` or ``?`` followed by
white space, which agrees with punkt on single-sentence quotes and ordinary sentence ends.  Everything else follows the
reference including its quirks: a count word after ``0`` is dropped ("0개" -> "영"), a decimal part in front of a count word
is dropped, a leading ``+`` makes the reference raise (``int('+')``) and so does this module.
"""
import ast
import json
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "ko_dictionary.json"), encoding="utf-8") as _f:
    _tables = json.load(_f)
etc_dictionary = dict(_tables["etc_dictionary"])
english_dictionary = dict(_tables["english_dictionary"])

# (13일)  and  (漢字): CJK radicals, Kangxi radicals, ideographic marks, Extension A, unified ideographs.  (The reference's
# class also lists three ranges that, as its source file is encoded, start and end on UNIFIED ideographs inside
# U+4E00-U+9FC3 - U+8C48-U+9DB4, U+4FAE-U+983B, U+4E26-U+9F8E - so the compatibility block U+F900.. is NOT matched; the
# class below accepts exactly the same code points, checked over the whole BMP.)
_DAY_NOTE = re.compile(r"\(\d+일\)")
_HANJA_NOTE = re.compile("\\([\u2e80-\u2e99\u2e9b-\u2ef3\u2f00-\u2fd5\u3005\u3007\u3021-\u3029\u3038-\u303a\u303b"
                         "\u3400-\u4db5\u4e00-\u9fc3]+\\)")
_QUOTED = re.compile("([`\"'\uff02\u201c\u2018])(.+?)([`\"'\uff02\u201d\u2019])")
_WORD = re.compile("[A-Za-z]+")

_LETTER_NAMES = dict(zip("ABCDEFGHIJKLMNOPQRSTUVWXYZ",
                         "에이 비 씨 디 이 에프 지 에이치 아이 제이 케이 엘 엠 엔 오 피 큐 알 에스 티 유 브이 더블유 엑스 와이 지".split()))
_UNITS_FIRST = {"%": "퍼센트", "cm": "센치미터", "mm": "밀리미터", "km": "킬로미터", "kg": "킬로그람"}
_UNITS_SECOND = {"m": "미터"}
_DIGIT_NAMES = dict(zip("0123456789", "영일이삼사오육칠팔구"))
_SINO = [""] + list("일이삼사오육칠팔구")                     # 1..9
_NATIVE = ["", "한", "두", "세", "네", "다섯", "여섯", "일곱", "여덟", "아홉"]
_PLACE = ["", "십", "백", "천"]                               # within a group of four digits
_GROUP = ["", "만", "억", "조", "경", "해"]                    # 10^4, 10^8, ...
_NATIVE_TENS = {"십": "열", "두십": "스물", "세십": "서른", "네십": "마흔", "다섯십": "쉰", "여섯십": "예순",
                "일곱십": "일흔", "여덟십": "여든", "아홉십": "아흔"}
_NATIVE_TENS_RE = re.compile("|".join(_NATIVE_TENS))          # in this order: "십" first, as the reference lists them
_NUMBER = r"([+-]?\d[\d,]*)[\.]?\d*"
_COUNT_WORD = "(시|명|가지|살|마리|포기|송이|수|톨|통|점|개|벌|척|채|다발|그루|자루|줄|켤레|그릇|잔|마디|상자|사람|곡|병|판)"
_NUMBER_WITH_COUNT = re.compile(_NUMBER + _COUNT_WORD)
_NUMBER_ALONE = re.compile(_NUMBER)


def _apply_table(text, table):
    """Replace every occurrence of a key of ``table``; ONE left-to-right pass, earlier keys win at a position."""
    if not any(key in text for key in table):
        return text
    return re.compile("|".join(re.escape(k) for k in table)).sub(lambda m: table[m.group()], text)


def _spell_capitals(m):
    word = m.group(0)
    if all(ch.isupper() for ch in word):
        return "".join(_LETTER_NAMES[ch] for ch in word)
    return word


def _split_sentences(text):
    parts = [p for p in re.split(r"(?<=[.!?])\s+", text.strip()) if p]
    return parts or [text]


def _requote(m):
    inner = m.group()[1:-1]
    return " ".join("'{}'".format(s) for s in _split_sentences(inner))


def _name_integer(digits, names):
    """Digit string -> numeral words, read in blocks of four decimal places (천 백 십 + unit digit, then 만 / 억 / 조 ...).

    Every digit gets the power of ten it stands for, counted from the length of the VALUE (``str(int(digits))``) - the
    reference derives its place names that way (text/korean.py:283-300), and so "007.5" names its digits three places too low.
    A block is spoken only if its units place (power % 4 == 0) is inside the string; its name is ``_GROUP[power // 4]`` with the
    list's own index rules (negative wraps, too large raises), which again is what the reference's table lookup does."""
    top = len(str(int(digits))) - 1
    powers = range(top, top - len(digits), -1)
    blocks = {}
    for ch, p in zip(digits, powers):
        if ch != "0":
            blocks.setdefault(p // 4, []).append(names[int(ch)] + _PLACE[p % 4])
    lowest = powers[-1] if len(digits) else 0
    return "".join("".join(blocks[b]) + _GROUP[b] for b in sorted(blocks, reverse=True) if 4 * b >= lowest)


def number_to_korean(m, is_count=False):
    """One number (a regex match) -> Korean words (reference text/korean.py:256-325).  ``is_count``: the number stands in front of a
    count word (group 2, kept): native numerals, and only group 1 - sign, digits, commas - is read, so a decimal part is lost."""
    literal = (m.group(1) if is_count else m.group()).replace(",", "")
    unit = m.group(2) if is_count else ""
    if ast.literal_eval(literal) == 0:              # literal_eval as the reference: a leading zero on an integer is a SyntaxError
        return "영"                                 # (the count word is lost with it)
    whole, dot, fraction = literal.partition(".")
    sign = {"-": "마이너스 ", "+": "플러스 "}.get(whole[0], "")
    if whole[0] == "+":
        int("+")                                    # the reference converts the string character by character: ValueError
    if whole[0] == "-":
        whole = str(abs(int(whole)))
    words = _name_integer(whole, _NATIVE if is_count else _SINO)
    leading_one = "한" if is_count else "일"        # 일십 -> 십, 한백 -> 백, but a bare 일 / 한 stays
    if len(words) > 1 and words.startswith(leading_one):
        words = words[1:]
    if is_count:
        words = _NATIVE_TENS_RE.sub(lambda t: _NATIVE_TENS[t.group()], words)
    if dot:
        words += "쩜 " + "".join(_DIGIT_NAMES[d] for d in fraction)
    return sign + words + unit


def normalize_number(text):
    text = _apply_table(text, _UNITS_FIRST)
    text = _apply_table(text, _UNITS_SECOND)
    text = _NUMBER_WITH_COUNT.sub(lambda m: number_to_korean(m, True), text)
    return _NUMBER_ALONE.sub(lambda m: number_to_korean(m, False), text)


def normalize(text):
    """Reference text/korean.py:163-177."""
    text = text.strip()
    text = _DAY_NOTE.sub("", text)
    text = _HANJA_NOTE.sub("", text)
    text = _apply_table(text, etc_dictionary)
    text = _WORD.sub(lambda m: english_dictionary.get(m.group(), m.group()), text)
    text = _WORD.sub(_spell_capitals, text)
    text = _QUOTED.sub(_requote, text)
    return normalize_number(text)
