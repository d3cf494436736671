"""The library's test hooks live in ONE environment variable, WO_TEST_HOOKS="key=value,key=value" (csrc/host_util.h: test_hook), read where the
product reads its options (once per API call / flood call).  set_hook / del_hook edit that variable through pytest's monkeypatch."""
import os


def _parse():
    cur = os.environ.get("WO_TEST_HOOKS", "")
    return dict(item.split("=", 1) if "=" in item else (item, "1") for item in cur.split(",") if item)


def _store(monkeypatch, d):
    if d:
        monkeypatch.setenv("WO_TEST_HOOKS", ",".join(f"{k}={v}" for k, v in d.items()))
    else:
        monkeypatch.delenv("WO_TEST_HOOKS", raising=False)


def set_hook(monkeypatch, key, value=1):
    d = _parse()
    d[key] = str(value)
    _store(monkeypatch, d)


def del_hook(monkeypatch, key):
    d = _parse()
    d.pop(key, None)
    _store(monkeypatch, d)
