"""Stand-in for airfoils.fileio (see package docstring): the .dat path is not pinned."""


def import_airfoil_data(filename):
    raise NotImplementedError("airfoils stand-in: .dat import is not available (parity unpinned)")
