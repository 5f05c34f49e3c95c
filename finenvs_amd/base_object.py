"""print() helper every reference object offers (finenvs/base_object.py:7-9)."""
from pprint import pprint


class BaseObject(object):
    def print(self) -> None:
        pprint(vars(self))
