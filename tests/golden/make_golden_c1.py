"""Golden fixture for BASELINE.json configs[0]: data/PrideandPrejudice.txt chunked, all-MiniLM-L6-v2
shaped encoder, HNSW, top-10 on CPU.

Run in the dev container only (needs /root/reference/data/PrideandPrejudice.txt):
    python tests/golden/make_golden_c1.py
Pipeline (all CPU): whitespace sliding-window chunks 256 words / 128 overlap (stand-in for llama-index'
SentenceSplitter(256,128) of apps/document_rag.py:42-47, which is not installed) -> stand-in WordPiece
tokenizer trained on the text (no pretrained vocab offline) -> BertEncoder fp32 (seeded random weights,
MiniLM-L6 architecture) -> host HNSW builder (M=32, efConstruction=200: hnsw_backend.py:54-55) ->
ORACLE search (ef=64, k=10, strict best-first).  Stored: packed token ids of chunks and queries, the
graph (compact-CSR file, written by our writer == the reference layout), and the oracle's top-10.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))

from leann_amd import csr_format as cf  # noqa: E402
from leann_amd.encoder import BertEncoder, config_for  # noqa: E402
from leann_amd.hnsw_builder import build_hnsw  # noqa: E402
from leann_amd.tokenizer import _train_wordpiece  # noqa: E402
from oracle import oracle as orc  # noqa: E402

TEXT = Path("/root/reference/data/PrideandPrejudice.txt")
QUERIES = [
    "Who does Elizabeth Bennet marry at the end of the story?",
    "What is Mr. Darcy's first impression of Elizabeth at the ball?",
    "Describe Mr. Collins and his proposal.",
    "Why does Lydia run away with Wickham?",
    "What does Lady Catherine de Bourgh demand of Elizabeth?",
    "Jane Bennet falls ill at Netherfield.",
    "Pemberley house and its grounds",
    "It is a truth universally acknowledged",
]


def main():
    words = TEXT.read_text(encoding="utf-8", errors="ignore").split()
    chunks = [" ".join(words[i : i + 256]) for i in range(0, max(1, len(words) - 128), 128)]
    tok = _train_wordpiece(chunks, 8192)
    tok.enable_truncation(max_length=256)
    enc_ids = [e.ids for e in tok.encode_batch(chunks)]
    q_ids = [e.ids for e in tok.encode_batch(QUERIES)]
    cfg = config_for("all-MiniLM-L6-v2")
    enc = BertEncoder.random_init(cfg, seed=0)

    def embed(seqs):
        T = max(len(s) for s in seqs)
        ids = torch.zeros((len(seqs), T), dtype=torch.int32)
        for i, s in enumerate(seqs):
            ids[i, : len(s)] = torch.tensor(s, dtype=torch.int32)
        lens = torch.tensor([len(s) for s in seqs], dtype=torch.int32)
        with torch.no_grad():
            return enc.encode_tokens(ids, lens, batch_size=64).numpy()

    X = embed(enc_ids)
    Q = embed(q_ids)
    g = build_hnsw(X, "mips", M=32, ef_construction=200, seed=7, num_threads=1)
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 384)
    ids, dist, st = orc.search(og, Q, 10, ef=64, beam=1, table=X)
    gt, _ = orc.bruteforce_topk(X, Q, 10, 0)
    rec = np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(QUERIES))])
    print(f"{len(chunks)} chunks, recall@10 of the oracle search vs exact: {rec:.3f}, ndis/query {st['ndis'] / len(QUERIES):.0f}")
    cf.write_index(HERE / "c1_pp.index", g, prune_embeddings=True)

    def pack(seqs):
        off = np.zeros(len(seqs) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        return np.concatenate([np.asarray(s, np.uint16) for s in seqs]), off

    ct, co = pack(enc_ids)
    qt, qo = pack(q_ids)
    np.savez_compressed(HERE / "c1_pp.npz", chunk_tok=ct, chunk_off=co, query_tok=qt, query_off=qo, oracle_ids=ids,
                        oracle_dist=dist, exact_ids=gt, query_emb=Q.astype(np.float32))
    print("wrote", (HERE / "c1_pp.npz").stat().st_size, (HERE / "c1_pp.index").stat().st_size)


if __name__ == "__main__":
    main()
