from efficient_gnns_amd.nn import GCNConv, SAGEConv  # noqa: F401
