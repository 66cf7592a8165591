CALLS = []


def log_scalar(key, value, weight=1, priority=10, round=None):
    CALLS.append(["scalar", key, float(value), float(weight), priority, round])


def log_derived(key, fn, priority=20):
    CALLS.append(["derived", key, priority])
