"""``--key v1 v2 ...`` command-line table with ``--arg_file`` support (behaviour of
``PARC/motion_tracker/util/arg_parser.py:15-133``)."""
import re


class ArgParser:
    def __init__(self):
        self._table = dict()

    def load_args(self, arg_strs):
        vals, key = [], ""
        for s in arg_strs:
            if s.startswith("#"):
                continue
            if s.startswith("--"):
                if key != "" and key not in self._table:
                    self._table[key] = vals
                vals, key = [], s[2:]
            else:
                vals.append(s)
        if key != "" and key not in self._table:
            self._table[key] = vals
        return True

    def load_file(self, filename):
        with open(filename, "r") as f:
            lines = re.split(r"[\n\r]+", f.read())
        strs = []
        for line in lines:
            if len(line) > 0 and not line.startswith("#"):
                strs += line.split()
        return self.load_args(strs)

    def has_key(self, key):
        return key in self._table

    def parse_string(self, key, default=""):
        return self._table[key][0] if self.has_key(key) else default

    def parse_strings(self, key, default=()):
        return list(self._table[key]) if self.has_key(key) else list(default)

    def parse_int(self, key, default=0):
        return int(self._table[key][0]) if self.has_key(key) else default

    def parse_ints(self, key, default=()):
        return [int(v) for v in self._table[key]] if self.has_key(key) else list(default)

    def parse_float(self, key, default=0.0):
        return float(self._table[key][0]) if self.has_key(key) else default

    def parse_floats(self, key, default=()):
        return [float(v) for v in self._table[key]] if self.has_key(key) else list(default)

    def parse_bool(self, key, default=False):
        if not self.has_key(key):
            return default
        return self._table[key][0].lower() in ("true", "1", "t", "y", "yes")
