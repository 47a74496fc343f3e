"""Text-table logger (behaviour of ``PARC/util/logger.py:68-142``): ``log(key, val, collection, quiet)``,
``print_log()`` to stdout, ``write_log()`` appends a tab-separated row.  wandb is optional and off by default."""
import os


class Logger:
    class Entry:
        def __init__(self, val, quiet=False, collection=None):
            self.val = val
            self.quiet = quiet
            self.collection = collection

    @staticmethod
    def print(s, end=None):
        print(s, end=end)

    def __init__(self):
        self.output_file = None
        self.log_headers = []
        self.log_current_row = {}
        self._row_count = 0
        self._step_key = None

    def set_step_key(self, key):
        self._step_key = key

    def configure_output_file(self, filename=None):
        self.log_headers = []
        self.log_current_row = {}
        self._row_count = 0
        if filename:
            d = os.path.dirname(filename)
            if d:
                os.makedirs(d, exist_ok=True)
            self.output_file = open(filename, "w")

    def log(self, key, val, collection=None, quiet=False):
        if key not in self.log_headers and self._row_count == 0:
            self.log_headers.append(key)
        self.log_current_row[key] = Logger.Entry(val, quiet, collection)

    def get_num_keys(self):
        return len(self.log_headers)

    def print_log(self):
        keys = [k for k in self.log_headers if k in self.log_current_row and not self.log_current_row[k].quiet]
        if not keys:
            return
        w = max(len(k) for k in keys)
        print("-" * (w + 22))
        for k in keys:
            v = self.log_current_row[k].val
            vs = "%8.4g" % v if isinstance(v, float) else str(v)
            print("| %*s | %15s |" % (w, k, vs))
        print("-" * (w + 22))

    def write_log(self):
        if self.output_file is not None:
            if self._row_count == 0:
                self.output_file.write("\t".join("{:<25s}".format(k) for k in self.log_headers) + "\n")
            vals = []
            for k in self.log_headers:
                e = self.log_current_row.get(k)
                vals.append("{:<25s}".format(str(e.val if e is not None else "")))
            self.output_file.write("\t".join(vals) + "\n")
            self.output_file.flush()
        self._row_count += 1
