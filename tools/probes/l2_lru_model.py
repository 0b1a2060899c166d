"""Host model (no GPU): LRU cache of 32 768 lines (one XCD's 4 MiB L2 at 128 B per source node and slice) fed with the
gather stream of the K = 256 aggregation in row order, for three node orders.  Would relabelling the nodes help?  (tools only)"""
import sys, numpy as np, time
from collections import OrderedDict
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from efficient_gnns_amd.data import powerlaw_edges, ARXIV
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
n=ARXIV['num_nodes']
ei=powerlaw_edges(n, ARXIV['num_edges'], max_degree=ARXIV['max_degree'], seed=0)
r=np.concatenate([ei[0],ei[1],np.arange(n)]); c=np.concatenate([ei[1],ei[0],np.arange(n)])
A=sp.csr_matrix((np.ones(len(r),np.int8),(r,c)),shape=(n,n)); A.sum_duplicates()
def hit_rate(A, cap=32768):
    # the short-row kernel walks rows in list order; many waves run concurrently, so the stream is roughly row-major
    indptr, idx = A.indptr, A.indices
    lru=OrderedDict(); hits=0
    for x in idx.tolist():
        if x in lru:
            hits+=1; lru.move_to_end(x)
        else:
            lru[x]=None
            if len(lru)>cap: lru.popitem(last=False)
    return hits/len(idx)
t=time.time(); print("natural order      L2-line LRU(32768) hit rate: %.3f" % hit_rate(A), "(%.0fs)"%(time.time()-t))
deg=np.asarray(A.sum(1)).ravel()
p=np.argsort(-deg, kind='stable'); P=A[p][:,p].tocsr(); print("degree-sorted      hit rate: %.3f" % hit_rate(P))
p=reverse_cuthill_mckee(A, symmetric_mode=True); P=A[p][:,p].tocsr(); print("reverse Cuthill-McKee hit rate: %.3f" % hit_rate(P))
for cap in (65536, 131072): print("natural, cache %d lines: %.3f" % (cap, hit_rate(A, cap)))
