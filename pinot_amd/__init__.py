"""pinot_amd — MI355X-native segment query executor for Apache Pinot's filter → projection → group-by hot path."""
__version__ = "0.1.0"
