def item(x):
    return x.item() if hasattr(x, "item") else x


def get_perplexity(loss, round=2, base=2):
    import builtins
    return 0.0 if loss is None else builtins.round(base ** loss, round)
