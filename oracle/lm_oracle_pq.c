/* PQ/DiskANN-style oracle: added with the PQ path */
int orc_pq_placeholder(void) { return 0; }
