class DescrStatsW:   # plotting path of associaTR only (hidden options); not provided
    def __init__(self, *a, **k):
        raise NotImplementedError("statsmodels stand-in: DescrStatsW is not provided")
