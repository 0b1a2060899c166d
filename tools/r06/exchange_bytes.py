"""Bytes a rank sends per epoch for the aggregations of the sharded step, in both exchange forms, at 2 / 4 / 8 ranks -- integer
arithmetic on the partition plan (dist.halo_rows_per_rank), no GPU.  GCN-256 on the arxiv-shaped graph: hidden aggregations K = 256
(forward, backward, eval), class-wide K = 40 (forward, backward, eval), the input layer's halo is static.  SAGE-256 (mean) on the
MAG-shaped graph: 6 aggregations of K = 256 per epoch (layers 2-3 forward, backward, eval; SAGEConv aggregates its input).
usage: python tools/r06/exchange_bytes.py [--mag-scale 1.0]"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import efficient_gnns_amd.data as D
import efficient_gnns_amd.dist as DD

ap = argparse.ArgumentParser()
ap.add_argument("--mag-scale", type=float, default=1.0)
ap.add_argument("--skip-mag", action="store_true")
args = ap.parse_args()


def table(name, adj, n, widths):
    rowptr, col, _ = adj.csr()
    out = {}
    for G in (2, 4, 8):
        halo = DD.halo_rows_per_rank(rowptr, col, n, G)
        mean_halo = sum(halo) / G
        row = dict(halo_rows_per_rank=halo, mean_halo_rows=round(mean_halo), sliced_row_equivalent=round(2 * n * (G - 1) / G ** 2))
        for K, calls in widths:
            halo_b = mean_halo * K * 4 * calls
            sliced_b = 2 * n * K * 4 * (G - 1) / G ** 2 * calls
            pays = K % (4 * G) == 0 and mean_halo * G * G > 2 * n * (G - 1)
            row[f"K={K} x{calls}"] = dict(halo_MB=round(halo_b / 1e6, 1), sliced_MB=round(sliced_b / 1e6, 1), auto="sliced" if pays else "halo")
        out[f"{G} ranks"] = row
    print(json.dumps({name: out}, indent=1))


d = D.arxiv_like(1.0, seed=0)
table("arxiv-shaped (Chung-Lu, N=%d): GCN-256, per epoch = train step + eval" % d.num_nodes, d.adj_t, d.num_nodes, [(256, 3), (40, 3)])
d2 = D.arxiv_like(1.0, seed=0, graph="local")
perm = None
table("community graph (ids shuffled, ranges as given)", d2.adj_t, d2.num_nodes, [(256, 3), (40, 3)])
if not args.skip_mag:
    m = DD.mag_problem(args.mag_scale, 0)
    # SAGEConv aggregates its INPUT (out = 349 > in = 256: no narrow-first form): layers 2 and 3 gather 256-wide hidden rows forward, backward
    # and in eval = 6 calls per epoch; layer 1's 128-wide input is static (halo fetched once, no input gradient)
    table("MAG-shaped (N=%d): SAGE-256 mean, 3 layers; per epoch 6 aggregations of K=256 (layers 2-3: forward, backward, eval)" % m.num_nodes,
          m.adj_t, m.num_nodes, [(256, 6)])
