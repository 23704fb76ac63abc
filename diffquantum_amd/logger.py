"""Text logger with the interface of the reference's ``Logger`` (/root/reference/logger.py:4-33):
``Logger(name=None, path=None)``, ``write_text(txt, silent=False)``, ``write_text_aux(txt, silent=True)``;
files go to ``<path>/text/<name>_<YYYYmmdd-HHMMSS>.txt`` (+ ``_aux``).  Unlike the reference the
directory is created on demand (the reference ships a pre-created ``logs/text``)."""
import os
import time


class Logger:
    def __init__(self, name=None, path=None):
        root = path if path is not None else os.path.join(os.getcwd(), "logs")
        self.dt_string = time.strftime("%Y%m%d-%H%M%S")
        stem = self.dt_string if name is None else "{}_{}".format(name, self.dt_string)
        os.makedirs(os.path.join(root, "text"), exist_ok=True)
        self.log_file = "{}/text/{}.txt".format(root, stem)
        self.log_file_aux = "{}/text/{}_aux.txt".format(root, stem)
        print("logs are written to {}".format(self.log_file))

    @staticmethod
    def _append(fname, txt, silent):
        with open(fname, "a") as f:
            f.write("{}\n".format(txt))
        if not silent:
            print(txt)

    def write_text(self, txt, silent=False):
        self._append(self.log_file, txt, silent)

    def write_text_aux(self, txt, silent=True):
        self._append(self.log_file_aux, txt, silent)
