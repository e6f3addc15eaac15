"""import-time placeholder"""
