"""Names of the node types and relations of the customer-article graph.  `Constants` exposes the attribute names the
reference's utils/constants.py:12-21 does (they are dictionary keys of every batch and of `state_dict`); the
module-level names the reference also exports are kept as aliases at the bottom."""
from typing import Tuple

EdgeKey = Tuple[str, str, str]


def _relation(source: str, name: str, target: str) -> EdgeKey:
    return (source, name, target)


class Constants:
    # node types
    node_user: str = "customer"
    node_item: str = "article"
    node_extra: str = "colour_group_code"   # the article column that can become a node type (data/types.py:25)
    # relation names
    rel_type: str = "buys"
    rel_rev_type: str = "rev_" + rel_type
    rel_type_extra: str = "has_color"
    # (source type, relation, destination type) keys of HeteroData
    edge_key: EdgeKey = _relation(node_user, rel_type, node_item)
    rev_edge_key: EdgeKey = _relation(node_item, rel_rev_type, node_user)
    edge_key_extra: EdgeKey = _relation(node_item, rel_type_extra, node_extra)


node_user, node_item, node_extra = Constants.node_user, Constants.node_item, Constants.node_extra
rel_type, rel_rev_type, rel_type_extra = Constants.rel_type, Constants.rel_rev_type, Constants.rel_type_extra
