"""Text -> token ids for the token store and for query strings.

The reference tokenises inside sentence-transformers (``model.encode``,
leann/embedding_compute.py:229-239; manual path ``tokenizer(..., truncation=True,
max_length=512)`` :299-305).  Here passages are tokenised ONCE at load.  Order of preference:
  1. the model's own Hugging Face tokenizer if it is available locally (offline cache / directory);
  2. a stand-in WordPiece tokenizer saved next to the index (``<index>.tokenizer.json``);
  3. a stand-in WordPiece tokenizer trained deterministically on the passages themselves and saved
     as (2) -- only reached when no pretrained vocabulary exists offline (random-weight encoders do
     not care which vocabulary is used).
"""

from __future__ import annotations

from pathlib import Path
from typing import Iterable, Optional


class TextTokenizer:
    def __init__(self, tok, max_len: int, kind: str):
        self._tok = tok
        self.max_len = max_len
        self.kind = kind
        self._tok.enable_truncation(max_length=max_len)
        self._tok.no_padding()

    @property
    def vocab_size(self) -> int:
        return self._tok.get_vocab_size()

    def encode_batch(self, texts: list[str]) -> list[list[int]]:
        return [e.ids for e in self._tok.encode_batch(list(texts))]


def _train_wordpiece(texts: Iterable[str], vocab_size: int):
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors, trainers

    tok = Tokenizer(models.WordPiece(unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    trainer = trainers.WordPieceTrainer(vocab_size=vocab_size, special_tokens=["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"],
                                        show_progress=False)
    tok.train_from_iterator(texts, trainer)
    tok.post_processor = processors.TemplateProcessing(
        single="[CLS] $A [SEP]", special_tokens=[("[CLS]", tok.token_to_id("[CLS]")), ("[SEP]", tok.token_to_id("[SEP]"))])
    return tok


def load_tokenizer(model_name: str, max_len: int, index_path: Optional[str] = None,
                   train_texts: Optional[Iterable[str]] = None, vocab_size: int = 30522) -> TextTokenizer:
    from tokenizers import Tokenizer

    # 1. the model's own tokenizer, if present locally
    try:
        p = Path(model_name)
        cand = p / "tokenizer.json" if p.is_dir() else None
        if cand is not None and cand.exists():
            return TextTokenizer(Tokenizer.from_file(str(cand)), max_len, "hf-local")
        from huggingface_hub import try_to_load_from_cache

        f = try_to_load_from_cache(model_name, "tokenizer.json")
        if isinstance(f, str):
            return TextTokenizer(Tokenizer.from_file(f), max_len, "hf-cache")
    except Exception:  # noqa: BLE001
        pass
    # 2. saved stand-in
    saved = Path(str(index_path) + ".tokenizer.json") if index_path else None
    if saved is not None and saved.exists():
        return TextTokenizer(Tokenizer.from_file(str(saved)), max_len, "stand-in")
    # 3. train a stand-in
    if train_texts is None:
        raise FileNotFoundError(
            f"no tokenizer for '{model_name}' available offline and no passages given to train a stand-in vocabulary")
    tok = _train_wordpiece(train_texts, vocab_size)
    if saved is not None:
        try:
            tok.save(str(saved))
        except OSError:
            pass
    return TextTokenizer(tok, max_len, "stand-in-trained")
