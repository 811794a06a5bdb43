"""`python -m leann_backend_diskann.diskann_embedding_server` -> the MI355X embedding server, protobuf wire format
(replaces packages/leann-backend-diskann/leann_backend_diskann/diskann_embedding_server.py; same command line, :435-462)."""
from leann_amd.embedding_server import main

if __name__ == "__main__":
    main(flavour="diskann")
