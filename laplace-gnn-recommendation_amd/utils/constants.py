"""Node / relation names of the customer-article graph (reference: utils/constants.py:4-21)."""

node_user = "customer"
node_item = "article"
rel_type = "buys"
rel_rev_type = "rev_buys"
node_extra = "colour_group_code"  # ArticleColumn.ColourGroupCode.value (data/types.py:25)
rel_type_extra = "has_color"


class Constants:
    node_user = node_user
    node_item = node_item
    node_extra = node_extra
    rel_type = rel_type
    rel_rev_type = rel_rev_type
    rel_type_extra = rel_type_extra
    edge_key = (node_user, rel_type, node_item)
    rev_edge_key = (node_item, rel_rev_type, node_user)
    edge_key_extra = (node_item, rel_type_extra, node_extra)
