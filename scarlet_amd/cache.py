"""Process-wide cache for expensive set-up products (monotonicity tables,
Fourier phase ramps), keyed by (name, key) like the reference's ``Cache``."""


class Cache:
    _cache = {}

    @staticmethod
    def check(name, key):
        return Cache._cache.setdefault(name, {})[key]

    @staticmethod
    def set(name, key, content):
        Cache._cache.setdefault(name, {})[key] = content
