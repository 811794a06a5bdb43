"""Text -> token ids for the token store and for query strings.

The reference tokenises inside sentence-transformers (``model.encode``,
leann/embedding_compute.py:229-239; manual path ``tokenizer(..., truncation=True,
max_length=512)`` :299-305).  Here passages are tokenised ONCE at load.  Order of preference:
  1. the model's own Hugging Face tokenizer if it is available locally (tokenizer.json, vocab.txt, or what
     AutoTokenizer can build offline);
  -- only with ``allow_stand_in`` (the encoder runs seeded RANDOM weights: benchmark / tests) --
  2. a stand-in WordPiece tokenizer saved next to the index (``<index>.tokenizer.json``);
  3. a stand-in WordPiece tokenizer trained deterministically on the passages themselves and saved as (2).
With real checkpoint weights a missing vocabulary is an error, never a stand-in.
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


def _from_vocab_txt(vocab_file: Path, lowercase: bool = True):
    """BERT WordPiece pipeline around a checkpoint's ``vocab.txt`` (checkpoints that ship no tokenizer.json)."""
    from tokenizers import Tokenizer, decoders, models, normalizers, pre_tokenizers, processors

    tok = Tokenizer(models.WordPiece.from_file(str(vocab_file), unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=lowercase)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.decoder = decoders.WordPiece()
    tok.post_processor = processors.TemplateProcessing(
        single="[CLS] $A [SEP]", special_tokens=[("[CLS]", tok.token_to_id("[CLS]")), ("[SEP]", tok.token_to_id("[SEP]"))])
    return tok


def _pretrained(model_name: str):
    """(tokenizers.Tokenizer, kind) of the model's OWN vocabulary if it is available locally, else None."""
    from tokenizers import Tokenizer

    p = Path(model_name)
    if p.is_dir():
        if (p / "tokenizer.json").exists():
            return Tokenizer.from_file(str(p / "tokenizer.json")), "hf-local"
        if (p / "vocab.txt").exists():
            lower = True
            try:
                import json

                lower = bool(json.loads((p / "tokenizer_config.json").read_text()).get("do_lower_case", True))
            except (OSError, ValueError):
                pass
            return _from_vocab_txt(p / "vocab.txt", lower), "hf-local-vocab"
        return None
    try:
        from huggingface_hub import try_to_load_from_cache

        f = try_to_load_from_cache(model_name, "tokenizer.json")
        if isinstance(f, str):
            return Tokenizer.from_file(f), "hf-cache"
        f = try_to_load_from_cache(model_name, "vocab.txt")
        if isinstance(f, str):
            return _from_vocab_txt(Path(f)), "hf-cache-vocab"
    except Exception:  # noqa: BLE001 - no hub cache in this environment
        pass
    try:  # slow-tokenizer-only checkpoints: let transformers convert (offline)
        from transformers import AutoTokenizer

        t = AutoTokenizer.from_pretrained(model_name, local_files_only=True, use_fast=True)
        if getattr(t, "backend_tokenizer", None) is not None:
            return t.backend_tokenizer, "hf-auto"
    except Exception:  # noqa: BLE001
        pass
    return None


def load_tokenizer(model_name: str, max_len: int, index_path: Optional[str] = None,
                   train_texts: Optional[Iterable[str]] = None, vocab_size: int = 30522,
                   allow_stand_in: bool = False) -> TextTokenizer:
    """The model's own tokenizer (tokenizer.json, vocab.txt or AutoTokenizer, all offline).  ``allow_stand_in`` --
    set ONLY when the encoder runs seeded random weights (synthetic benchmark, tests: the vocabulary is then
    arbitrary) -- additionally permits a stand-in WordPiece vocabulary saved next to the index or trained on the
    passages.  With real checkpoint weights a missing vocabulary raises: token ids from any other vocabulary index
    the wrong rows of the embedding table and the search silently returns garbage.  ``vocab_size`` = the encoder's
    embedding-table rows; a tokenizer that can emit larger ids is rejected."""
    from tokenizers import Tokenizer

    got = _pretrained(model_name)
    if got is not None:
        tt = TextTokenizer(got[0], max_len, got[1])
        if tt.vocab_size > vocab_size:
            raise ValueError(f"tokenizer of {model_name!r} has {tt.vocab_size} entries but the encoder's embedding table has {vocab_size} rows")
        return tt
    if not allow_stand_in:
        raise FileNotFoundError(
            f"no tokenizer (tokenizer.json / vocab.txt) for {model_name!r} available offline; a stand-in vocabulary is only "
            "permitted with random-weight encoders (allow_stand_in=True)")
    saved = Path(str(index_path) + ".tokenizer.json") if index_path else None
    if saved is not None and saved.exists():
        tt = TextTokenizer(Tokenizer.from_file(str(saved)), max_len, "stand-in")
        if tt.vocab_size > vocab_size:
            raise ValueError(f"saved stand-in tokenizer has {tt.vocab_size} entries > embedding table rows {vocab_size}")
        return tt
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
