from efficient_gnns_amd.utils import softmax, subgraph  # noqa: F401


def _not_on_hot_path(name):
    def f(*a, **k):
        raise NotImplementedError(f"torch_geometric.utils.{name} is imported but never called by the reference's hot path")
    return f


to_dense_adj = _not_on_hot_path("to_dense_adj")
negative_sampling = _not_on_hot_path("negative_sampling")
add_self_loops = _not_on_hot_path("add_self_loops")
