from efficient_gnns_amd.nn import GATConv, GCNConv, SAGEConv  # noqa: F401
