"""Host-side mirror of the reference's ``utils`` package for the rule-editing hot path
(same module and symbol names as /root/reference/utils)."""
