"""Part of the test-tooling tf stand-in (see ../__init__.py): train.py imports tf_debug and never uses it."""
debug = None
