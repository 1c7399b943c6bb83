class FPN:  # never constructed by the golden generator
    pass


class LastLevelMaxPool:
    pass


class LastLevelP6P7:
    pass
