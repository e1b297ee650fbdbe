"""Host-side mirror of SALIENT++'s ``fast_trainer`` data-path modules (samplers, transferers,
monkeypatch, shufflers) on top of the MI355X ``fast_sampler`` replacement."""
