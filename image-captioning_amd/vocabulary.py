"""Vocabulary as the generate path uses it (dataset/vocabulary.py:17-38,91-136,176-188).

Only what `generate` touches is mirrored: the four special ids, ``str_to_idx``, ``idx_to_str``,
``encode_as_words`` (ids -> caption string), ``word_counts`` and ``len``.  Corpus tokenisation and
vocabulary building are data preparation and stay with the reference.
"""
from __future__ import annotations

from collections import Counter
from typing import Optional

from .detok import treebank_detokenize

PAD, SOS, EOS, UNK = "<PAD>", "<SOS>", "<EOS>", "<UNK>"  # constants.py:39-42


class Vocabulary:
    def __init__(self, tokenizer: str, freq_threshold: Optional[int] = None, text=None, sp_model_path: Optional[str] = None):
        if tokenizer not in ("word", "sp-bpe"):
            raise ValueError("Invalid tokenizer type.")
        self.tokenizer = tokenizer
        self.freq_threshold = freq_threshold
        self.stoi_dict = {PAD: 0, SOS: 1, EOS: 2, UNK: 3}
        self.itos_dict = {0: PAD, 1: SOS, 2: EOS, 3: UNK}
        self.word_counts = Counter()
        self.sp = None
        if text is not None:
            raise NotImplementedError("building a vocabulary from text is data preparation; load a state dict instead")
        if tokenizer == "sp-bpe" and sp_model_path is not None:
            import sentencepiece as spm
            self.sp = spm.SentencePieceProcessor()
            self.sp.load(sp_model_path)

    @classmethod
    def synthetic(cls, size: int) -> "Vocabulary":
        """Word vocabulary w4..w{size-1} with counts (i % 97) + 1, for benchmarks and tests."""
        v = cls("word")
        for i in range(4, size):
            w = f"w{i}"
            v.stoi_dict[w] = i
            v.itos_dict[i] = w
            v.word_counts[w] = (i % 97) + 1
        v.word_counts[SOS] = 50
        v.word_counts[EOS] = 50
        return v

    def counts_by_id(self):
        """Corpus count of every id, the input of the output layer's frequency bias."""
        out = [0.0] * len(self)
        for w, c in self.word_counts.items():
            out[self.str_to_idx(w)] = float(c)
        return out

    def str_to_idx(self, word: str) -> int:
        if self.tokenizer == "word":
            return self.stoi_dict.get(word, self.stoi_dict[UNK])
        return self.sp.piece_to_id(word)

    def idx_to_str(self, idx: int) -> str:
        if self.tokenizer == "word":
            return self.itos_dict.get(idx, UNK)
        return self.sp.id_to_piece(idx)

    def encode_as_words(self, idxs) -> str:
        """ids -> text (dataset/vocabulary.py:106-121).  Word mode drops ids 0, 1, 2 and detokenises the words the way NLTK's
        TreebankWordDetokenizer does (detok.py restates its rules: NLTK is not installed here)."""
        if self.tokenizer == "word":
            return treebank_detokenize(self.idx_to_str(int(i)) for i in idxs if int(i) not in (0, 1, 2))
        return self.sp.decode_ids([int(i) for i in idxs])

    def load_dict(self, state_dict: dict) -> None:
        if self.tokenizer != "word":
            raise ValueError("Invalid tokenizer type.")
        self.stoi_dict = state_dict["str_to_idx"]
        self.itos_dict = state_dict["idx_to_str"]
        self.word_counts = state_dict["word_counts"]
        self.freq_threshold = state_dict["freq_threshold"]

    def get_state_dict(self) -> dict:
        if self.tokenizer != "word":
            raise ValueError("Invalid tokenizer type.")
        return {"str_to_idx": self.stoi_dict, "idx_to_str": self.itos_dict, "word_counts": self.word_counts,
                "freq_threshold": self.freq_threshold}

    def __len__(self) -> int:
        if self.tokenizer == "word":
            return len(self.stoi_dict)
        return self.sp.get_piece_size()
