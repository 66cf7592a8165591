def safe_round(number, ndigits):
    return round(number.item() if hasattr(number, "item") else number, ndigits)
