#!/usr/bin/env python3
"""Generate the text front-end fixtures by running the REFERENCE's normalize() (text/korean.py:163-326).

Runs only in the build container (needs /root/reference).  The reference is imported read-only with
PYTHONDONTWRITEBYTECODE=1; third-party modules it imports but that are absent here (jamo, nltk, unidecode, inflect) are
replaced by EMPTY stub modules - normalize() needs none of them except nltk.sent_tokenize inside quoted text, so inputs with
quotes fail under the stubs and are recorded as unpinned (no expected output is written for them).

Writes
  * text2speech_amd/text/ko_dictionary.json : the two lookup tables of text/ko_dictionary.py as DATA (the dict objects the
    reference builds at import: insertion order kept, duplicate keys already resolved the way Python resolves them);
  * tests/golden/text_normalize.json        : [{"text": ..., "normalized": ...}] = the reference's outputs.
"""
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

for name in ("jamo", "nltk", "unidecode", "inflect"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["jamo"].hangul_to_jamo = sys.modules["jamo"].h2j = sys.modules["jamo"].j2h = None      # names only; never called here
sys.modules["unidecode"].unidecode = None
sys.modules["inflect"].engine = lambda: None          # text/en_numbers.py builds an engine at import; normalize() never uses it
sys.path.insert(0, "/root/reference")

from text import korean as ref_korean  # noqa: E402  (the reference)
from text import ko_dictionary as ref_dict  # noqa: E402

INPUTS = [
    # the reference's own demo inputs (text/korean.py:335-341)
    "JTBC는 JTBCs를 DY는 A가 Absolute",
    "오늘(13일) 3,600마리 강아지가",
    "60.3%",
    '"저돌"(猪突) 입니다.',
    "비대위원장이 지난 1월 이런 말을 했습니다. “난 그냥 산돼지처럼 돌파하는 스타일이다”",
    "지금은 -12.35%였고 종류는 5가지와 19가지, 그리고 55가지였다",
    "JTBC는 TH와 K 양이 2017년 9월 12일 오후 12시에 24살이 된다",
    # plain Hangul: untouched
    "존경하는 국민 여러분",
    "안녕하세요, 반갑습니다!",
    # digits: plain, with separators, signs, decimals, zero, large
    "1", "10", "11", "21", "100", "101", "110", "1000", "1001", "10000", "12345", "100000", "1234567", "100000000",
    "123456789012", "0", "0.5", "3.14", "+5", "-7", "-0.25", "1,000", "12,345,678", "2017년", "9월 12일",
    "가격은 15000원입니다", "온도는 -3.5도", "1억 2천만", "제 3의 물결", "5.18 민주화운동",
    # counters (native numerals)
    "1명", "2명", "3개", "4시", "5가지", "10개", "11마리", "12시", "20살", "21살", "30명", "45개", "99병", "100개",
    "101명", "24살이", "3,600마리", "7송이", "19그루", "55가지", "1사람", "10명과 20개",
    # units
    "5%", "10cm", "3mm", "42km", "70kg", "100m", "10m와 5cm", "50% 할인", "키는 180cm 몸무게는 75kg",
    # English words / abbreviations
    "KTX", "LG", "CNN 뉴스", "IT 산업", "PC방", "TV", "DVD와 CCTV", "idol", "track", "up", "down", "Devsisters",
    "ABC", "XYZ", "NASA", "Hello", "iPhone", "MIT와 KAIST", "BBC는 UN", "A B C", "GDP는 3% 성장", "DNA", "CEO가 FTA",
    "Q and A", "OK", "SNS에서", "IMF 외환위기", "UFC 200", "AI 시대", "K리그", "Y S", "francisco", "humble apology",
    # dictionary entries of etc_dictionary
    "20~30대", "2 30대", "20, 30대", "1+1", "3에서 6개월인",
    # parenthesised dates / hanja
    "어제(12일) 그리고 오늘(13일)", "중국(中國)과 한국(韓國)", "(1일)", "사과(沙果) 3개",
    # mixed
    "2017년 9월 12일 오후 12시에 KTX를 타고 100km를 갔다", "JTBC 뉴스룸 8시", "코스피 2,400선 3.5% 상승",
    "LA에서 5명", "10% 20% 30%", "1,2,3", "3-4", "A4 용지 500장", "V3", "  앞뒤 공백 1개  ",
    # number corners (round 3): trailing dot, zeros in front of a decimal point, every block name, zeros inside blocks, signs
    # with count words, counts >= 100, a rejected literal (leading zero on an integer)
    "3.", "7.0", "00.5", "05.5", "007.5", "00012345.5", "0000012.5", "10.05", "1000.001", "-0", "0.0", "-0.0",
    "10001", "10010", "100100", "1000000", "10000000", "1000000000000", "10000000000000000", "100000000000000000000",
    "99999999", "100020003", "9007199254740993", "110개", "111명", "1000명", "1234개", "10000개", "20000마리", "-3개", "-12.5개",
    "1,234.5", "12,34", "3.14.15", "1.2개", "90살", "60세 70살 80명", "007", "1000000000000000000000000",
]


def main():
    # tables as data
    tables = {"etc_dictionary": list(ref_dict.etc_dictionary.items()),
              "english_dictionary": list(ref_dict.english_dictionary.items())}
    p = os.path.join(ROOT, "text2speech_amd", "text", "ko_dictionary.json")
    with open(p, "w", encoding="utf-8") as f:
        json.dump(tables, f, ensure_ascii=False, indent=0)
    out, unpinned = [], []
    for text in INPUTS:
        try:
            out.append({"text": text, "normalized": ref_korean.normalize(text)})
        except Exception as e:      # noqa: BLE001  (quotes need nltk.sent_tokenize, absent here)
            unpinned.append({"text": text, "why": "%s: %s" % (type(e).__name__, e)})
    p = os.path.join(ROOT, "tests", "golden", "text_normalize.json")
    with open(p, "w", encoding="utf-8") as f:
        json.dump({"cases": out, "unpinned": unpinned}, f, ensure_ascii=False, indent=0)
    print("pinned %d cases, %d unpinned" % (len(out), len(unpinned)))
    for u in unpinned:
        print("  unpinned:", u["text"], "|", u["why"])


if __name__ == "__main__":
    main()
