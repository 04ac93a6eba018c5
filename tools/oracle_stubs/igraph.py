"""Stand-in for python-igraph (absent offline): the reference only needs the name."""


class Graph:
    pass
