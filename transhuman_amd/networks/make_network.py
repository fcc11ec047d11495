"""make_network -- /root/reference/lib/networks/make_network.py:4-11."""
from .cross_transformer import Network


def make_network(cfg=None):
    return Network()
