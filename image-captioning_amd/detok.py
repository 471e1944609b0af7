"""Treebank detokenisation for word-mode captions (dataset/vocabulary.py:113-117 calls NLTK's
``TreebankWordDetokenizer().detokenize(words)``).

NLTK is not installed where this package runs, so this is a restatement of that class's published algorithm
(nltk/tokenize/treebank.py, unchanged from NLTK 3.5 through 3.9): join with spaces, pad with one space on each side, then
apply its substitution lists in its order — the MacIntyre contractions, ending quotes, double dashes, brackets,
punctuation, starting quotes — and strip.  **Parity unpinned**: the reference holds no detokenised caption to compare
with; the tests hold the examples of NLTK's own documentation.
"""
from __future__ import annotations

import re
from typing import Iterable

# MacIntyreContractions, with the tokenizer's "(?#X)" marker replaced by the white space the tokenizer inserted
_CONTRACTIONS2 = [re.compile(p.replace("(?#X)", r"\s")) for p in (
    r"(?i)\b(can)(?#X)(not)\b", r"(?i)\b(d)(?#X)('ye)\b", r"(?i)\b(gim)(?#X)(me)\b", r"(?i)\b(gon)(?#X)(na)\b",
    r"(?i)\b(got)(?#X)(ta)\b", r"(?i)\b(lem)(?#X)(me)\b", r"(?i)\b(more)(?#X)('n)\b", r"(?i)\b(wan)(?#X)(na)(?=\s)")]
_CONTRACTIONS3 = [re.compile(p.replace("(?#X)", r"\s")) for p in (r"(?i) ('t)(?#X)(is)\b", r"(?i) ('t)(?#X)(was)\b")]

_ENDING_QUOTES = [
    (re.compile(r"([^' ])\s('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) "), r"\1\2 "),
    (re.compile(r"([^' ])\s('[sS]|'[mM]|'[dD]|') "), r"\1\2 "),
    (re.compile(r"(\S)\s(\'\')"), r"\1\2"),
    (re.compile(r"(\'\')\s([.,:)\]>};%])"), r"\1\2"),
    (re.compile(r"''"), '"'),
]
_DOUBLE_DASHES = (re.compile(r" -- "), r"--")
_PARENS_BRACKETS = [
    (re.compile(r"([\[\(\{\<])\s"), r"\g<1>"),
    (re.compile(r"\s([\]\)\}\>])"), r"\g<1>"),
    (re.compile(r"([\]\)\}\>])\s([:;,.])"), r"\1\2"),
]
_PUNCTUATION = [
    (re.compile(r"([^'])\s'\s"), r"\1' "),
    (re.compile(r"\s([?!])"), r"\g<1>"),
    (re.compile(r'([^\.])\s(\.)([\]\)}>"\']*)\s*$'), r"\1\2\3"),
    (re.compile(r"([#$])\s"), r"\g<1>"),
    (re.compile(r"\s([;%])"), r"\g<1>"),
    (re.compile(r"\s\.\.\.\s"), r"..."),
    (re.compile(r"\s([:,])"), r"\1"),
]
_STARTING_QUOTES = [
    (re.compile(r"([ (\[{<])\s``"), r"\1``"),
    (re.compile(r"(``)\s"), r"\1"),
    (re.compile(r"``"), r'"'),
]


def treebank_detokenize(tokens: Iterable[str]) -> str:
    text = " " + " ".join(tokens) + " "
    for rx in _CONTRACTIONS3:
        text = rx.sub(r"\1\2", text)
    for rx in _CONTRACTIONS2:
        text = rx.sub(r"\1\2", text)
    for rx, sub in _ENDING_QUOTES:
        text = rx.sub(sub, text)
    text = text.strip()
    text = _DOUBLE_DASHES[0].sub(_DOUBLE_DASHES[1], text)
    for rx, sub in _PARENS_BRACKETS:
        text = rx.sub(sub, text)
    for rx, sub in _PUNCTUATION:
        text = rx.sub(sub, text)
    for rx, sub in _STARTING_QUOTES:
        text = rx.sub(sub, text)
    return text.strip()
