"""import-only stub"""
