"""stand-in sub-package (see ../README.md)"""
