from efficient_gnns_amd.transforms import ToSparseTensor  # noqa: F401
