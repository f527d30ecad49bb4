from six import with_metaclass  # noqa: F401
