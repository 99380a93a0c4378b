__version__ = "0.1.0"
__reference_version__ = "2.2.0"  # krypy API version mirrored by this package
