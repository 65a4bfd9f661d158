for a in "" "--mesh 60x40"; do ./examples/level_pipeline 300 $a | tail -3; done
