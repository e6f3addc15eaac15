class Average:
    pass


class MeanStd:
    pass


class PearsonCorrelation:
    pass
