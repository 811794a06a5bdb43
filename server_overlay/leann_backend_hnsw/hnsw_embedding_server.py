"""`python -m leann_backend_hnsw.hnsw_embedding_server` -> the MI355X embedding / distance server, msgpack wire format
(replaces packages/leann-backend-hnsw/leann_backend_hnsw/hnsw_embedding_server.py; same command line, :395-417)."""
from leann_amd.embedding_server import main

if __name__ == "__main__":
    main(flavour="hnsw")
